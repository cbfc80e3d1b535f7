#!/usr/bin/env python3
"""Writes tests/golden/pin_vectors.bin: what the CPU oracle (oracle/) says real OpenCV + the reference's unmodified
src/ORBextractor.cc must produce on two seeded frames, stage by stage -- the expectations tools/pin_against_opencv/pin_orb.cpp
checks on a machine that HAS OpenCV (this container has none: every parity statement of this repository stays "unpinned" until
that program has been run).  Records: name[48] | dtype u32 (0 u8, 1 i32, 2 f32, 3 u32) | count u64 | payload.

  usage: python tests/golden/make_pin_vectors.py [out]      (tests/test_pin_kit.py checks the committed file against a re-run)"""
import hashlib
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))
import oracle  # noqa: E402
import spec_numpy as spec  # noqa: E402
from orbhip import synth  # noqa: E402

FRAMES = [(0, 640, 480), (1, 752, 480)]            # (generator index, width, height); nFeatures 1000, 8 levels, 1.2, 20 / 7
PRESETS = [(18, 34, 49, 55), (18, 34, 48, 56)]     # orb_gaussian_preset 0 / 1
DT = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2, np.dtype(np.uint32): 3}


def row_hashes(img):
    h = np.full(img.shape[0], 2166136261, np.uint64)
    for x in range(img.shape[1]):
        h = ((h ^ img[:, x].astype(np.uint64)) * np.uint64(16777619)) & np.uint64(0xFFFFFFFF)
    return h.astype(np.uint32)


def atan_inputs():
    g = np.arange(-40, 41, dtype=np.float32)
    yy, xx = np.meshgrid(g, g, indexing="ij")
    return yy.ravel(), xx.ravel()


def build():
    recs = []
    add = lambda name, arr: recs.append((name, np.ascontiguousarray(arr)))
    ys, xs = atan_inputs()
    add("atan2.deg", np.array([oracle.fast_atan2(float(y), float(x)) for y, x in zip(ys, xs)], np.float32))
    for fi, (idx, w, h) in enumerate(FRAMES):
        img = synth.synth_frame(idx, w, h)
        ref = oracle.Extractor(1000, 1.2, 8, 20, 7)
        kps, desc = ref.extract(img)
        add("f%d.dims" % fi, np.array([idx, w, h, len(kps)], np.int32))
        add("f%d.sha256" % fi, np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8))
        for l in range(8):
            lv = ref.pyramid_level(l)
            add("f%d.pyr%d.dims" % (fi, l), np.array([lv.shape[1], lv.shape[0]], np.int32))
            add("f%d.pyr%d.rows" % (fi, l), row_hashes(lv))
        lv0 = ref.pyramid_level(0)
        for p, taps in enumerate(PRESETS):
            add("f%d.blur0.p%d.rows" % (fi, p), row_hashes(oracle.blur(lv0, taps)))
        # cv::FAST(level 0, threshold 20, nonmaxSuppression = true) on the WHOLE image: (x, y, response) ascending y, x
        add("f%d.fast0" % fi, spec._fast_cell(lv0, 20).astype(np.int32))
        # per level: what ComputeKeyPointsOctTree leaves in allKeypoints[level] (x, y with the +16 border offset, response, angle)
        t = ref.tables()
        for l in range(8):
            lv = ref.pyramid_level(l)
            c = ref.cell_candidates(l)
            k = ref.distribute(c, 16, lv.shape[1] - 16, 16, lv.shape[0] - 16, int(t["quota"][l]))
            out = np.zeros((len(k), 4), np.float32)
            for i, (x, y, r) in enumerate(k):
                out[i] = (x + 16, y + 16, r, ref.ic_angle(l, int(x) + 16, int(y) + 16))
            add("f%d.oct%d" % (fi, l), out)
        add("f%d.kps" % fi, np.frombuffer(kps.tobytes(), np.uint8))
        add("f%d.desc" % fi, desc)
    return recs


def write(path):
    with open(path, "wb") as f:
        for name, arr in build():
            f.write(name.encode().ljust(48, b"\0"))
            f.write(struct.pack("<IQ", DT[arr.dtype], arr.size))
            f.write(arr.tobytes())


if __name__ == "__main__":
    write(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "pin_vectors.bin"))
