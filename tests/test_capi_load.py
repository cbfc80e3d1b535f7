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


def test_abi_version_and_struct_sizes(lib):
    """orb_featstore grew; callers can compare the library's idea of it with their own."""
    hdr = open(os.path.join(ROOT, "include", "orb_hip.h")).read()
    assert lib.orb_abi_version() == int(re.search(r"#define ORB_HIP_ABI_VERSION (\d+)", hdr).group(1))
    lib.orb_sizeof_featstore.restype = ctypes.c_size_t
    assert lib.orb_sizeof_featstore() == ctypes.sizeof(capi.FeatStoreC)


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


def test_shard_range_is_a_partition_and_matches_the_python_rule():
    """orb_shard_range (the C ABI's frame partition of orb_multi_extract_batch) covers every frame exactly once, in
    order, with blocks that differ by at most one frame, and is the rule bench.py / orbhip.shard use."""
    from orbhip import shard
    for total in (0, 1, 7, 64, 512, 513, 1000):
        for world in (1, 2, 3, 4, 8, 16):
            nxt, sizes = 0, []
            for r in range(world):
                first, count = capi.shard_range(total, world, r)
                assert (first, count) == shard.frame_range(total, world, r)
                assert first == nxt and count >= 0
                nxt += count
                sizes.append(count)
            assert nxt == total and max(sizes) - min(sizes) <= 1
    assert capi.shard_range(10, 0, 0) == (0, 0) and capi.shard_range(10, 4, 4) == (0, 0)


def test_multi_without_a_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.OrbError):
        capi.MultiExtractor([0, 1])


def test_c_caller_of_the_abi_compiles_and_links(lib):
    """tools/latency_c.c is a plain-C caller of include/orb_hip.h (the header must stay C-clean, and every entry point it
    uses must link from liborbhip.so): built here with gcc, run only on the GPU box (tools/latency_c.sh)."""
    import subprocess
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "orb-slam2-chinesenotes_amd"), "latency-c"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.exists(os.path.join(ROOT, "tools", "latency_c"))
