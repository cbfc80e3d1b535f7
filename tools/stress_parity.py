#!/usr/bin/env python3
"""One-off parity stress on the GPU box (not part of the suites): many seeded random SearchByBoW cases built to make the
features of a vocabulary node COMPETE for the same partner (the greedy "already taken" rule, the row-coupling test of the
4-row kernel path, nodes of 1..40 features), both variants, against the oracle.   usage: stress_parity.py [trials]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (one HIP runtime per process: torch first)
import oracle  # noqa: E402
from orbhip import capi, synth  # noqa: E402


def fv(desc):
    f = oracle.bow_transform(desc, synth.synth_vocabulary())
    return f, (f.node_ids, f.offsets, f.indices)


def competing_sets(rng, na, nb, reuse, maxflip):
    db = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
    if nb == 0:
        return rng.integers(0, 256, (na, 32), dtype=np.uint8), db
    pool = rng.integers(0, nb, max(1, int(nb * reuse)))            # few B features are everybody's partner
    src = pool[rng.integers(0, len(pool), na)]
    da = db[src].copy()
    for r in range(na):
        for b in rng.integers(0, 256, rng.integers(0, maxflip + 1)):
            da[r, b >> 3] ^= np.uint8(1 << (b & 7))
    return da, db


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(20261004)
    matches = 0
    for t in range(trials):
        na, nb = int(rng.integers(1, 1600)), int(rng.integers(1, 1600))
        if t % 7 == 0:
            na, nb = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        reuse, maxflip = float(rng.choice([0.05, 0.2, 0.6, 1.0])), int(rng.choice([2, 8, 30, 60]))
        ratio, ori = float(rng.choice([0.6, 0.7, 0.9, 1.0])), bool(rng.integers(0, 2))
        da, db = competing_sets(rng, na, nb, reuse, maxflip)
        aa = rng.uniform(0, 360, na).astype(np.float32)
        ab = rng.uniform(0, 360, nb).astype(np.float32)
        va = (rng.random(na) < 0.8).astype(np.uint8)
        vb = (rng.random(nb) < 0.8).astype(np.uint8)
        fva, ta = fv(da)
        fvb, tb = fv(db)
        m = capi.Matcher(ratio, ori)
        wn, w = oracle.search_by_bow(da, aa, va, fva, db, ab, fvb, ratio, ori)
        gn, g = m.search_by_bow(da, aa, va, ta, db, ab, tb)
        assert gn == wn and np.array_equal(g, w), ("KF-F", t, na, nb, reuse, maxflip, ratio, ori)
        wn2, w2 = oracle.search_by_bow_kk(da, aa, va, fva, db, ab, vb, fvb, ratio, ori)
        gn2, g2 = m.search_by_bow_kk(da, aa, va, ta, db, ab, vb, tb)
        assert gn2 == wn2 and np.array_equal(g2, w2), ("KF-KF", t, na, nb, reuse, maxflip, ratio, ori)
        matches += wn + wn2
        m.close()
        if t % 25 == 24:
            print("trial %d ok, %d matches so far" % (t + 1, matches), flush=True)
    print("stress_parity: %d trials identical to the oracle, %d matches" % (trials, matches))


if __name__ == "__main__":
    main()
