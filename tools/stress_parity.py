#!/usr/bin/env python3
"""One-off parity stress on the GPU box (not part of the suites): many seeded random SearchByBoW cases built to make the
features of a vocabulary node COMPETE for the same partner (the greedy "already taken" rule, the row-coupling test of the
4-row kernel path, nodes of 1..40 features), both variants, against the oracle.   usage: stress_parity.py [trials] | --extract [trials] | --query [trials]"""
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


def random_image(rng, w, h):
    kind = int(rng.integers(0, 6))
    if kind == 0:                                                  # the benchmark's scenes
        return synth.synth_frame(int(rng.integers(0, 100000)), w, h)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 1:                                                  # blocks of random grey: corners everywhere
        p = int(rng.integers(3, 12))
        return np.kron(rng.integers(0, 256, (h // p + 1, w // p + 1)), np.ones((p, p)))[:h, :w].astype(np.uint8)
    if kind == 2:                                                  # corners clustered in one or two spots, the rest flat
        img = np.full((h, w), int(rng.integers(60, 200)), np.uint8)
        for _ in range(int(rng.integers(1, 3))):
            bw, bh = int(rng.integers(20, max(21, w // 3))), int(rng.integers(20, max(21, h // 3)))
            x0, y0 = int(rng.integers(0, w - bw)), int(rng.integers(0, h - bh))
            img[y0:y0 + bh, x0:x0 + bw] = np.kron(rng.integers(0, 256, (bh // 5 + 1, bw // 5 + 1)), np.ones((5, 5)))[:bh, :bw]
        return img
    if kind == 3:                                                  # noise: every pixel a candidate at the low threshold
        return rng.integers(0, 256, (h, w)).astype(np.uint8)
    if kind == 4:                                                  # gradient + mild noise: few, weak corners (minTh fallback)
        return np.clip(xx * 255 // w + rng.integers(-9, 10, (h, w)), 0, 255).astype(np.uint8)
    return (((yy // int(rng.integers(1, 9)) + xx // int(rng.integers(1, 9))) & 1) * int(rng.integers(40, 220)) + 20).astype(np.uint8)


def extract_stress(trials):
    """Random image sizes, feature counts, pyramid shapes and contents through orb_extract against the oracle, byte for
    byte (exercises the quadtree's bucket / rank sorts and their fallbacks, strip overflow, tiny upper levels)."""
    rng = np.random.default_rng(777)
    total = 0
    for t in range(trials):
        w, h = int(rng.integers(96, 900)), int(rng.integers(96, 700))
        nf = int(rng.choice([50, 200, 500, 1000, 2000, 3500]))
        levels, sf = int(rng.choice([3, 5, 8, 8, 8])), float(rng.choice([1.2, 1.2, 1.1, 1.5]))
        ini, mn = int(rng.choice([20, 20, 12, 40])), int(rng.choice([7, 7, 5]))
        img = np.ascontiguousarray(random_image(rng, w, h))
        try:
            ex = capi.Extractor(nf, sf, levels, ini, mn)
            kps, desc = ex.extract(img)
        except capi.OrbError as e:
            if e.code == -5:                                       # ORB_ERR_UNSUPPORTED: outside the geometry envelope
                continue
            raise
        ref = oracle.Extractor(nf, sf, levels, ini, mn)
        rk, rd = ref.extract(img)
        assert kps.tobytes() == rk.tobytes() and np.array_equal(desc, rd), ("extract", t, w, h, nf, levels, sf, ini, mn)
        total += len(kps)
        ex.close()
        if t % 25 == 24:
            print("extract trial %d ok, %d keypoints so far" % (t + 1, total), flush=True)
    print("stress_parity --extract: %d trials identical to the oracle, %d keypoints" % (trials, total))


def query_stress(trials):
    """orb_match_bow_query_device (one query against many keyframes) on random feature stores: node counts, node-size skew, reuse
    of partners, capacities, ratios; every pair against the oracle (tests/test_gpu_matcher_query.py holds the store builder)."""
    import test_gpu_matcher_query as tq
    rng = np.random.default_rng(4096)
    total = pairs = 0
    for t in range(trials):
        n_nodes = int(rng.choice([8, 40, 100, 100, 250]))
        w = rng.random(n_nodes) ** float(rng.choice([0.5, 2.0, 6.0]))      # flat ... a few heavy nodes
        if t % 5 == 0:
            w[: int(rng.integers(1, 4))] = float(rng.choice([5.0, 20.0, 60.0])) * w.max()
        cap, n_kf, n_q = int(rng.choice([64, 300, 700, 1100])), int(rng.integers(1, 200)), int(rng.integers(1, 4))
        ratio, ori = float(rng.choice([0.6, 0.7, 0.9, 1.0])), bool(rng.integers(0, 2))
        desc, kps, valid, node, counts = tq._store(rng, n_kf, n_q, cap, n_nodes, w, reuse=float(rng.choice([0.05, 0.3, 1.0])),
                                                   maxflip=int(rng.choice([2, 12, 40])), p_valid=float(rng.choice([0.3, 0.7, 1.0])))
        mt = capi.Matcher(ratio, ori)
        kf_list = list(range(n_kf))
        q_list = list(range(n_kf, n_kf + n_q))
        m, n, m2, n2 = tq._run(mt, desc, kps, valid, node, counts, n_nodes, kf_list, q_list, True)
        assert np.array_equal(n, n2), ("query vs pair kernel", t)
        for qi, fq in enumerate(q_list):
            for ki in rng.choice(n_kf, min(n_kf, 25), replace=False):
                wn, wm = tq._oracle_pair(desc, kps, valid, node, counts, int(ki), fq, ratio, ori, True)
                nb = int(counts[fq])
                assert n[qi, ki] == wn and np.array_equal(m[qi, ki, :nb], wm), (t, qi, int(ki), n_nodes, cap, ratio, ori)
                total += wn
                pairs += 1
            assert np.array_equal(m[qi][:, :int(counts[fq])], m2[qi][:, :int(counts[fq])]), ("rows vs pair kernel", t, qi)
        mt.close()
        if t % 20 == 19:
            print("query trial %d ok, %d pairs vs oracle, %d matches so far" % (t + 1, pairs, total), flush=True)
    print("stress_parity --query: %d stores, %d pairs identical to the oracle (every pair identical to the pair kernel), %d matches" % (trials, pairs, total))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--extract":
        return extract_stress(int(sys.argv[2]) if len(sys.argv) > 2 else 150)
    if len(sys.argv) > 1 and sys.argv[1] == "--query":
        return query_stress(int(sys.argv[2]) if len(sys.argv) > 2 else 100)
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
