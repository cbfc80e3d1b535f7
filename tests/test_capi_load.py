"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, and exports every symbol that
include/orb_hip.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle
from orbhip import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    capi.build_library()
    return ctypes.CDLL(capi.LIB_PATH)


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "orb_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(orb_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_keypoint_layout_is_cv_keypoint():
    assert capi.KP_DTYPE.itemsize == 28 and capi.KP_DTYPE == oracle.KP_DTYPE


def test_builtin_pattern_matches_header():
    pat = capi.builtin_pattern()
    txt = open(os.path.join(ROOT, "include", "orb_brief_pattern.h")).read()
    vals = [int(v) for v in re.findall(r"-?\d+", txt.split("{", 1)[1].split("}", 1)[0])]
    assert pat.tolist() == vals


def test_host_hamming_and_three_maxima_match_oracle():
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        want = int(np.unpackbits(a ^ b).sum())
        assert capi.hamming(a, b) == want == oracle.hamming(a, b)
    for _ in range(300):
        counts = rng.integers(0, 12, 30).astype(np.int32)
        if rng.random() < 0.3:
            counts[rng.integers(0, 30)] = 200
        assert capi.three_maxima(counts) == oracle.three_maxima(counts)
    assert capi.three_maxima(np.zeros(30, np.int32)) == (-1, -1, -1)


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.OrbError) as e:
        capi.Extractor()
    assert e.value.code in (-3, -2)
