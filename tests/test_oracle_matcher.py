"""CPU tests of the matcher half of the oracle against brute-force Python restatements on small
inputs (the reference ships no fixtures for these; SURVEY Appendix B is the specification)."""
import numpy as np

import oracle
from orbhip import synth


def _rand_desc(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def test_hamming_vs_bitcount():
    rng = np.random.default_rng(0)
    for _ in range(100):
        a, b = _rand_desc(rng, 1)[0], _rand_desc(rng, 1)[0]
        want = sum(int(x).bit_count() for x in (a ^ b))
        assert oracle.hamming(a, b) == want
    z = np.zeros(32, np.uint8)
    assert oracle.hamming(z, z) == 0 and oracle.hamming(z, ~z) == 256


def test_three_maxima_rules():
    c = np.zeros(30, np.int32)
    assert oracle.three_maxima(c) == (-1, -1, -1)
    c[5] = 10
    assert oracle.three_maxima(c) == (5, -1, -1)
    c[7] = 10                     # tie: earlier bin stays first (strict >)
    assert oracle.three_maxima(c) == (5, 7, -1)
    c[2] = 1                      # 1 < 0.1*10 is false (1 < 1.0 false) -> third kept
    assert oracle.three_maxima(c) == (5, 7, 2)
    c[7] = 0                      # second = 1: 1 < 1.0 false -> kept; third 0 < 1 -> dropped
    assert oracle.three_maxima(c) == (5, 2, -1)
    c[5] = 11                     # 1 < 1.1 -> second and third dropped
    assert oracle.three_maxima(c) == (5, -1, -1)
    # note :1703 compares max3 with max1 (not max2)
    c[:] = 0
    c[0], c[1], c[2] = 100, 12, 9
    assert oracle.three_maxima(c) == (0, 1, -1)


def test_bow_transform_is_csr_of_first_min_descent():
    rng = np.random.default_rng(1)
    cent = synth.synth_vocabulary()
    desc = _rand_desc(rng, 200)
    fv = oracle.bow_transform(desc, cent)
    assert np.all(np.diff(fv.node_ids.astype(int)) > 0)
    assert fv.offsets[0] == 0 and fv.offsets[-1] == 200 and sorted(fv.indices.tolist()) == list(range(200))
    for k, nid in enumerate(fv.node_ids):
        idx = fv.indices[fv.offsets[k]:fv.offsets[k + 1]]
        assert np.all(np.diff(idx) > 0)
        for i in idx:
            d1 = [oracle.hamming(desc[i], cent[c]) for c in range(10)]
            c1 = int(np.argmin(d1))
            d2 = [oracle.hamming(desc[i], cent[10 + 10 * c1 + c]) for c in range(10)]
            assert nid == 11 + 10 * c1 + int(np.argmin(d2))


def _py_search_by_bow(dkf, akf, vkf, fvk, df, af, fvf, ratio, ori, kk=False, vf=None):
    """Literal Python transcription of SURVEY B.2 for cross-checking the C++ oracle."""
    import math
    nres = len(dkf) if kk else len(df)
    out = [-1] * nres
    taken = [False] * len(df)
    hist = [[] for _ in range(30)]
    nm = 0
    nodes_k = {int(n): fvk.indices[fvk.offsets[i]:fvk.offsets[i + 1]] for i, n in enumerate(fvk.node_ids)}
    nodes_f = {int(n): fvf.indices[fvf.offsets[i]:fvf.offsets[i + 1]] for i, n in enumerate(fvf.node_ids)}
    for nid in sorted(set(nodes_k) & set(nodes_f)):
        for ik in nodes_k[nid]:
            if not vkf[ik]:
                continue
            b1, b2, bi = 256, 256, -1
            for jf in nodes_f[nid]:
                if kk:
                    if taken[jf] or not vf[jf]:
                        continue
                elif out[jf] >= 0:
                    continue
                d = oracle.hamming(dkf[ik], df[jf])
                if d < b1:
                    b2, b1, bi = b1, d, jf
                elif d < b2:
                    b2 = d
            ok = (b1 < 50) if kk else (b1 <= 50)
            if ok and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                if kk:
                    out[ik] = bi
                    taken[bi] = True
                else:
                    out[bi] = ik
                rot = np.float32(akf[ik]) - np.float32(af[bi])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360))
                v = float(np.float32(rot * np.float32(1.0 / 30)))
                b = int(math.floor(v + 0.5))
                if b == 30:
                    b = 0
                if ori:
                    hist[b].append(ik if kk else bi)
                nm += 1
    if ori:
        i1, i2, i3 = oracle.three_maxima(np.array([len(h) for h in hist], np.int32))
        for b in range(30):
            if b in (i1, i2, i3):
                continue
            for j in hist[b]:
                out[j] = -1
                nm -= 1
    return nm, np.array(out, np.int32)


def test_search_by_bow_vs_python_transcription():
    rng = np.random.default_rng(2)
    cent = synth.synth_vocabulary()
    base = _rand_desc(rng, 300)
    flips = rng.integers(0, 256, base.shape, dtype=np.uint8) & rng.integers(0, 256, base.shape, dtype=np.uint8) & \
        rng.integers(0, 256, base.shape, dtype=np.uint8) & rng.integers(0, 256, base.shape, dtype=np.uint8)
    other = (base ^ flips)[rng.permutation(300)[:260]]
    akf = rng.uniform(0, 360, 300).astype(np.float32)
    af = rng.uniform(0, 360, 260).astype(np.float32)
    vkf = synth.synth_valid_flags(300, 5)
    vf = synth.synth_valid_flags(260, 6)
    fvk, fvf = oracle.bow_transform(base, cent), oracle.bow_transform(other, cent)
    for ratio, ori in [(0.7, True), (0.9, False)]:
        nm, out = oracle.search_by_bow(base, akf, vkf, fvk, other, af, fvf, ratio, ori)
        pn, pout = _py_search_by_bow(base, akf, vkf, fvk, other, af, fvf, ratio, ori)
        assert nm == pn and np.array_equal(out, pout)
        nm, out = oracle.search_by_bow_kk(base, akf, vkf, fvk, other, af, vf, fvf, ratio, ori)
        pn, pout = _py_search_by_bow(base, akf, vkf, fvk, other, af, fvf, ratio, ori, kk=True, vf=vf)
        assert nm == pn and np.array_equal(out, pout)


def test_grid_query_vs_brute_force():
    rng = np.random.default_rng(3)
    n = 500
    kps = np.zeros(n, oracle.KP_DTYPE)
    kps["x"] = rng.uniform(0, 640, n).astype(np.float32)
    kps["y"] = rng.uniform(0, 480, n).astype(np.float32)
    kps["octave"] = rng.integers(0, 3, n)
    grid = (0.0, 0.0, 0.1, 0.1)
    for _ in range(50):
        x, y, r = rng.uniform(-50, 700), rng.uniform(-50, 530), float(rng.choice([10, 30, 100]))
        got = oracle.features_in_area(kps, grid, x, y, r, 0, 0)
        def cell(v):
            return int(np.floor(np.float32(v) * np.float32(0.1) + np.float32(0.5)))
        # PosInGrid rounds, so keypoints in the last half cell fall outside the 64x48 grid and are in no cell
        # (src/Frame.cc:412-420): the reference never returns them, and neither does the brute force below
        want = [i for i in range(n) if kps["octave"][i] == 0 and abs(kps["x"][i] - np.float32(x)) < r
                and abs(kps["y"][i] - np.float32(y)) < r and cell(kps["x"][i]) < 64 and cell(kps["y"][i]) < 48]
        assert sorted(got.tolist()) == want        # same SET as brute force (cell ranges are conservative)
        # order: ascending cell column, then cell row, then index
        cells = [(int(np.floor(np.float32(kps["x"][i]) * np.float32(0.1) + np.float32(0.5))),
                  int(np.floor(np.float32(kps["y"][i]) * np.float32(0.1) + np.float32(0.5))), i) for i in got]
        assert cells == sorted(cells)


def test_search_for_init_identical_frames_matches_identity():
    ref = oracle.Extractor(2000, 1.2, 8, 20, 7)
    k, d = ref.extract(synth.synth_frame(3, 320, 240))
    prev = np.ascontiguousarray(np.stack([k["x"], k["y"]], axis=1), dtype=np.float32)
    nm, m12 = oracle.search_for_init(k, d, k, d, (0, 0, 64 / 320, 48 / 240), prev, 100, 0.9, True)
    lvl0 = k["octave"] == 0
    assert np.all(m12[~lvl0] == -1)
    hit = m12 >= 0
    assert nm == hit.sum() and nm > 0.5 * lvl0.sum()
    assert np.all(m12[hit] == np.nonzero(hit)[0])
