"""GPU parity tests of the matcher: HIP kernels through the C ABI vs the CPU oracle, bit-exact
match indices and counts."""
import numpy as np
import pytest

import oracle
from orbhip import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def feats():
    """Features of a few related frames (oracle-extracted: the matcher tests must not depend on the
    extractor kernels being right).  Frame k+100 is frame k with fresh noise -> many true matches."""
    ref = oracle.Extractor()
    out = []
    base = synth.synth_frame(0, noise=0).astype(np.int16)
    for s in range(4):
        nz = (synth.splitmix64(1234 + s, 0, base.size) % np.uint64(13)).astype(np.int16).reshape(base.shape) - 6
        img = np.clip(base + nz, 0, 255).astype(np.uint8)
        if s == 3:
            img = np.roll(img, 7, axis=1)
        out.append(ref.extract(img))
    return out


def _fv(desc):
    fv = oracle.bow_transform(desc, synth.synth_vocabulary())
    return fv, (fv.node_ids, fv.offsets, fv.indices)


@pytest.mark.parametrize("ratio,ori", [(0.7, True), (0.75, True), (0.9, False), (0.6, True)])
def test_search_by_bow_kf_frame(feats, ratio, ori):
    m = capi.Matcher(ratio, ori)
    for a, b in [(0, 1), (1, 2), (2, 3), (3, 0)]:
        (ka, da), (kb, db) = feats[a], feats[b]
        fva, ta = _fv(da)
        fvb, tb = _fv(db)
        valid = synth.synth_valid_flags(len(ka), a)
        want_n, want = oracle.search_by_bow(da, ka["angle"], valid, fva, db, kb["angle"], fvb, ratio, ori)
        got_n, got = m.search_by_bow(da, ka["angle"], valid, ta, db, kb["angle"], tb)
        assert want_n > 50
        assert got_n == want_n
        assert np.array_equal(got, want)


@pytest.mark.parametrize("ratio,ori", [(0.75, True), (0.8, False)])
def test_search_by_bow_kf_kf(feats, ratio, ori):
    m = capi.Matcher(ratio, ori)
    for a, b in [(0, 1), (2, 1), (3, 2)]:
        (ka, da), (kb, db) = feats[a], feats[b]
        fva, ta = _fv(da)
        fvb, tb = _fv(db)
        va = synth.synth_valid_flags(len(ka), 10 + a)
        vb = synth.synth_valid_flags(len(kb), 20 + b)
        want_n, want = oracle.search_by_bow_kk(da, ka["angle"], va, fva, db, kb["angle"], vb, fvb, ratio, ori)
        got_n, got = m.search_by_bow_kk(da, ka["angle"], va, ta, db, kb["angle"], vb, tb)
        assert want_n > 30
        assert got_n == want_n
        assert np.array_equal(got, want)


def test_bow_edge_cases(feats):
    m = capi.Matcher(0.7, True)
    (ka, da), (kb, db) = feats[0], feats[1]
    fva, ta = _fv(da)
    fvb, tb = _fv(db)
    # nothing valid on the keyframe side -> no matches
    n, out = m.search_by_bow(da, ka["angle"], np.zeros(len(ka), np.uint8), ta, db, kb["angle"], tb)
    assert n == 0 and np.all(out == -1)
    # empty frame
    e = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    n, out = m.search_by_bow(da, ka["angle"], np.ones(len(ka), np.uint8), ta, np.zeros((0, 32), np.uint8),
                             np.zeros(0, np.float32), e)
    assert n == 0 and len(out) == 0
    # disjoint vocabularies -> the merge walk never meets
    tb2 = (tb[0] + 1000, tb[1], tb[2])
    n, out = m.search_by_bow(da, ka["angle"], np.ones(len(ka), np.uint8), ta, db, kb["angle"], tb2)
    assert n == 0 and np.all(out == -1)
    # identical frames: every valid feature whose node-mates are not too similar matches itself
    valid = np.ones(len(ka), np.uint8)
    want_n, want = oracle.search_by_bow(da, ka["angle"], valid, fva, da, ka["angle"], fva, 0.7, True)
    n, out = m.search_by_bow(da, ka["angle"], valid, ta, da, ka["angle"], ta)
    assert n == want_n and np.array_equal(out, want)
    assert np.all(out[out >= 0] == np.nonzero(out >= 0)[0])


def test_one_big_node_more_than_64_features(feats):
    # all features in one vocabulary node: exercises the > 64-lane chunk loop and the greedy order
    m = capi.Matcher(0.8, True)
    (ka, da), (kb, db) = feats[0], feats[1]
    na, nb = 300, 333
    fa = (np.array([7], np.uint32), np.array([0, na], np.int32), np.arange(na, dtype=np.int32))
    fb = (np.array([7], np.uint32), np.array([0, nb], np.int32), np.arange(nb, dtype=np.int32))
    valid = synth.synth_valid_flags(na, 3)
    want_n, want = oracle.search_by_bow(da[:na], ka["angle"][:na], valid, oracle.FeatVec(*fa), db[:nb],
                                        kb["angle"][:nb], oracle.FeatVec(*fb), 0.8, True)
    n, out = m.search_by_bow(da[:na], ka["angle"][:na], valid, fa, db[:nb], kb["angle"][:nb], fb)
    assert want_n > 20
    assert n == want_n and np.array_equal(out, want)


@pytest.fixture(scope="module")
def feats2000():
    ref = oracle.Extractor(2000, 1.2, 8, 20, 7)         # Tracking.cc:121 builds the init extractor with 2x features
    base = synth.synth_frame(40, noise=0).astype(np.int16)
    out = []
    for s, shift in [(0, 0), (1, 5), (2, 23)]:
        nz = (synth.splitmix64(777 + s, 0, base.size) % np.uint64(13)).astype(np.int16).reshape(base.shape) - 6
        img = np.clip(np.roll(base, shift, axis=1) + nz, 0, 255).astype(np.uint8)
        out.append(ref.extract(img))
    return out


GRID_640 = (0.0, 0.0, 64.0 / 640.0, 48.0 / 480.0)


@pytest.mark.parametrize("window,ratio,ori", [(100, 0.9, True), (30, 0.9, True), (100, 0.8, False), (10, 0.9, True)])
def test_search_for_initialization(feats2000, window, ratio, ori):
    m = capi.Matcher(ratio, ori)
    for a, b in [(0, 1), (0, 2), (1, 0)]:
        (k1, d1), (k2, d2) = feats2000[a], feats2000[b]
        prev_ref = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1), dtype=np.float32)
        prev_gpu = prev_ref.copy()
        want_n, want = oracle.search_for_init(k1, d1, k2, d2, GRID_640, prev_ref, window, ratio, ori)
        got_n, got = m.search_for_initialization(k1, d1, k2, d2, GRID_640, prev_gpu, window)
        assert got_n == want_n
        assert np.array_equal(got, want)
        assert prev_gpu.tobytes() == prev_ref.tobytes()
        if window >= 30:
            assert want_n > 40
        # second call with the updated vbPrevMatched, as MonocularInitialization does on the next frame
        want_n2, want2 = oracle.search_for_init(k1, d1, k2, d2, GRID_640, prev_ref, window, ratio, ori)
        got_n2, got2 = m.search_for_initialization(k1, d1, k2, d2, GRID_640, prev_gpu, window)
        assert got_n2 == want_n2 and np.array_equal(got2, want2) and prev_gpu.tobytes() == prev_ref.tobytes()


def test_search_for_initialization_edges(feats2000):
    m = capi.Matcher(0.9, True)
    (k1, d1), (k2, d2) = feats2000[0], feats2000[1]
    prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1), dtype=np.float32)
    n, out = m.search_for_initialization(k1, d1, k2[:0], d2[:0], GRID_640, prev.copy(), 100)
    assert n == 0 and np.all(out == -1)
    # windows entirely outside the grid
    far = prev.copy() + 5000
    p2 = far.copy()
    wn, w = oracle.search_for_init(k1, d1, k2, d2, GRID_640, far, 100, 0.9, True)
    n, out = m.search_for_initialization(k1, d1, k2, d2, GRID_640, p2, 100)
    assert n == wn == 0 and np.array_equal(out, w)
    # a grid with an offset / different cell size (undistorted image bounds, Frame.cc:108-109)
    g = (-12.5, -7.25, 64.0 / 670.0, 48.0 / 495.0)
    pr, pg = prev.copy(), prev.copy()
    wn, w = oracle.search_for_init(k1, d1, k2, d2, g, pr, 60, 0.9, True)
    n, out = m.search_for_initialization(k1, d1, k2, d2, g, pg, 60)
    assert n == wn and np.array_equal(out, w) and pg.tobytes() == pr.tobytes()


def test_c5_style_stream_vs_keyframe_db_device_batch():
    """Config 5 in miniature: frames of a 752x480 stream, each matched by SearchByBoW against every keyframe of
    a descriptor DB that lives in HBM (the Relocalization candidate loop, Tracking.cc:1471-1492), through the
    device-resident batch entry points; every (keyframe, frame) pair bit-exact vs the oracle."""
    import torch
    nkf, nq, W, H = 12, 3, 752, 480
    frames = synth.synth_batch(300, nkf + nq, W, H)
    frames[nkf:] = frames[:nq]                                   # queries revisit keyframe scenes ...
    for q in range(nq):                                          # ... under fresh noise
        nz = (synth.splitmix64(4242 + q, 0, W * H) % np.uint64(9)).astype(np.int16).reshape(H, W) - 4
        frames[nkf + q] = np.clip(frames[q].astype(np.int16) + nz, 0, 255).astype(np.uint8)
    ex = capi.Extractor()
    mt = capi.Matcher(0.75, True)                                # Tracking.cc:1458
    cap, F = ex.max_keypoints, nkf + nq
    dev = torch.device("cuda", 0)
    d_imgs = torch.from_numpy(frames).to(dev)
    d_kps = torch.zeros(F * cap * 28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(F * cap * 32, dtype=torch.uint8, device=dev)
    d_counts = torch.zeros(F, dtype=torch.int32, device=dev)
    d_node = torch.zeros(F * cap, dtype=torch.int16, device=dev)
    valid_np = np.stack([synth.synth_valid_flags(cap, 900 + i) for i in range(F)])
    d_valid = torch.from_numpy(valid_np).to(dev)
    d_cent = torch.from_numpy(synth.synth_vocabulary()).to(dev)
    pairs = [(k, nkf + q) for q in range(nq) for k in range(nkf)]
    kf_idx = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=dev)
    f_idx = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=dev)
    d_match = torch.zeros(len(pairs) * cap, dtype=torch.int32, device=dev)
    d_nm = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ex.extract_batch_device(d_imgs.data_ptr(), F, H, W, W, W * H, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_counts.data_ptr())
    mt.wait_for(ex.stream)
    mt.bow_assign_device(d_desc.data_ptr(), d_counts.data_ptr(), F, cap, d_cent.data_ptr(), d_node.data_ptr())
    store = dict(desc=d_desc.data_ptr(), kps=d_kps.data_ptr(), valid=d_valid.data_ptr(), counts=d_counts.data_ptr(),
                 node_of=d_node.data_ptr(), cap=cap, n_frames=F)
    mt.match_bow_batch_device(store, kf_idx.data_ptr(), f_idx.data_ptr(), len(pairs), d_match.data_ptr(), d_nm.data_ptr())
    ex.sync(); mt.sync(); torch.cuda.synchronize()
    counts = d_counts.cpu().numpy()
    kps = d_kps.cpu().numpy().view(capi.KP_DTYPE).reshape(F, cap)
    desc = d_desc.cpu().numpy().reshape(F, cap, 32)
    match = d_match.cpu().numpy().reshape(len(pairs), cap)
    nm = d_nm.cpu().numpy()
    ref = oracle.Extractor()
    cent = synth.synth_vocabulary()
    feats = []
    for i in range(F):
        k, d = ref.extract(frames[i])
        assert counts[i] == len(k) and kps[i, :len(k)].tobytes() == k.tobytes() and np.array_equal(desc[i, :len(k)], d)
        feats.append((k, d, oracle.bow_transform(d, cent)))
    best = []
    for p, (a, b) in enumerate(pairs):
        (ka, da, fa), (kb, db, fb) = feats[a], feats[b]
        wn, wm = oracle.search_by_bow(da, ka["angle"], valid_np[a][:len(ka)], fa, db, kb["angle"], fb, 0.75, True)
        assert nm[p] == wn and np.array_equal(match[p, :len(kb)], wm), (a, b)
        best.append(wn)
    best = np.array(best).reshape(nq, nkf)
    assert np.all(best.argmax(axis=1) == np.arange(nq))          # each query recognises its own keyframe

    # the same batch with the per-frame feature vectors precomputed in the store (orb_bow_build_csr_device), plus two
    # invalid pairs (frame index outside the store / negative): reported as nmatches = -1, never run
    nn = 128
    d_ck = torch.zeros(F * cap, dtype=torch.int32, device=dev)
    d_cs = torch.zeros(F * nn, dtype=torch.int16, device=dev)
    d_cc = torch.zeros(F * nn, dtype=torch.int16, device=dev)
    mt.build_csr_device(d_node.data_ptr(), d_counts.data_ptr(), F, cap, nn, d_ck.data_ptr(), d_cs.data_ptr(), d_cc.data_ptr())
    store2 = dict(store, n_nodes=nn, csr_keys=d_ck.data_ptr(), csr_start=d_cs.data_ptr(), csr_cnt=d_cc.data_ptr())
    kf2 = torch.cat([kf_idx, torch.tensor([F + 3, 0], dtype=torch.int32, device=dev)])
    f2 = torch.cat([f_idx, torch.tensor([0, -1], dtype=torch.int32, device=dev)])
    d_match2 = torch.full(((len(pairs) + 2) * cap,), 7, dtype=torch.int32, device=dev)
    d_nm2 = torch.zeros(len(pairs) + 2, dtype=torch.int32, device=dev)
    mt.match_bow_batch_device(store2, kf2.data_ptr(), f2.data_ptr(), len(pairs) + 2, d_match2.data_ptr(), d_nm2.data_ptr())
    mt.sync(); torch.cuda.synchronize()
    nm2 = d_nm2.cpu().numpy()
    match2 = d_match2.cpu().numpy().reshape(len(pairs) + 2, cap)
    assert np.array_equal(nm2[:len(pairs)], nm) and nm2[-2] == -1 and nm2[-1] == -1
    for p, (a, b) in enumerate(pairs):
        assert np.array_equal(match2[p, :counts[b]], match[p, :counts[b]])
    assert np.all(match2[-2:] == -1)


def _proj_scene(feats, a, b, mode, level_mode, seed):
    """Queries = features of frame a 'projected' near their true position in frame b (same scene, fresh noise)."""
    rng = np.random.default_rng(seed)
    (ka, da), (kb, db) = feats[a], feats[b]
    nq, n = len(ka), len(kb)
    scale = np.float32(1.2) ** ka["octave"].astype(np.float32)
    q = np.zeros(nq, oracle.PROJ_DTYPE)
    q["x"] = ka["x"] + rng.uniform(-3, 3, nq).astype(np.float32)
    q["y"] = ka["y"] + rng.uniform(-3, 3, nq).astype(np.float32)
    lvl = ka["octave"]
    if mode == 0:
        q["r"] = np.float32(15.0) * scale                          # th * mvScaleFactors[nLastOctave]   (:215)
        if level_mode == "forward":
            q["min_level"], q["max_level"] = lvl, -1
        elif level_mode == "backward":
            q["min_level"], q["max_level"] = 0, lvl
        else:
            q["min_level"], q["max_level"] = lvl - 1, lvl + 1
        q["er_max"] = q["r"]
    else:
        r = np.where(rng.random(nq) < 0.5, np.float32(2.5), np.float32(4.0)) * np.float32(3.0)
        q["r"] = r.astype(np.float32) * scale                      # r * mvScaleFactors[nPredictedLevel]   (:97)
        q["min_level"], q["max_level"] = lvl - 1, lvl
        q["er_max"] = q["r"]
    q["ur"] = q["x"] - rng.uniform(5, 40, nq).astype(np.float32)
    q["flags"] = (rng.random(nq) < 0.85).astype(np.int32) | ((rng.random(nq) < 0.8).astype(np.int32) << 1)
    u_right = np.where(rng.random(n) < 0.5, kb["x"] - rng.uniform(5, 40, n), -1).astype(np.float32)
    occupied = (rng.random(n) < 0.1).astype(np.uint8)
    return q, da, ka["angle"].copy(), kb, db, u_right, occupied


@pytest.mark.parametrize("mode,level_mode", [(0, "neither"), (0, "forward"), (0, "backward"), (1, "map")])
def test_search_by_projection(feats, mode, level_mode):
    grid = (0.0, 0.0, 64.0 / 640.0, 48.0 / 480.0)
    for ori in (True, False):
        m = capi.Matcher(0.8, ori)
        for a, b, seed in [(0, 1, 1), (1, 2, 2), (2, 0, 3)]:
            q, qd, qa, kb, db, ur, occ = _proj_scene(feats, a, b, mode, level_mode, seed)
            want_n, want = oracle.search_by_projection(mode, q, qd, qa, kb, db, ur, occ, grid, 0.8, ori)
            got_n, got = m.search_by_projection(mode, q, qd, qa, kb, db, ur, occ, grid)
            assert want_n > 100
            assert got_n == want_n
            assert np.array_equal(got, want)


def test_search_by_projection_edges(feats):
    grid = (0.0, 0.0, 0.1, 0.1)
    m = capi.Matcher(0.8, True)
    q, qd, qa, kb, db, ur, occ = _proj_scene(feats, 0, 1, 0, "neither", 9)
    q["flags"] = 0                                                  # nothing live
    n, out = m.search_by_projection(0, q, qd, qa, kb, db, ur, occ, grid)
    assert n == 0 and np.all(out == -1)
    q, qd, qa, kb, db, ur, occ = _proj_scene(feats, 0, 1, 1, "map", 10)
    occ[:] = 1                                                      # every feature already has a MapPoint
    wn, w = oracle.search_by_projection(1, q, qd, qa, kb, db, ur, occ, grid, 0.8, True)
    n, out = m.search_by_projection(1, q, qd, qa, kb, db, ur, occ, grid)
    assert n == wn == 0 and np.array_equal(out, w)
    q["flags"] = 1                                                  # MapPoints without observations never block: features get reassigned
    occ[:] = 0
    wn, w = oracle.search_by_projection(0, q, qd, qa, kb, db, ur, occ, grid, 0.8, True)
    n, out = m.search_by_projection(0, q, qd, qa, kb, db, ur, occ, grid)
    assert n == wn and np.array_equal(out, w)


def test_distinctive_descriptors_batch():
    rng = np.random.default_rng(11)
    # (256 is the last size whose distance rows fit LDS at once; 257, 300, 513 and 1100 take the histogram form: ADVICE r4 --
    #  the reference handles any N, src/MapPoint.cc:306-335)
    sizes = [1, 2, 3, 4, 5, 8, 17, 33, 64, 65, 100, 200, 0, 7, 256, 257, 300, 513, 1100] + rng.integers(1, 40, 300).tolist()
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    # observations of one MapPoint resemble each other: a base descriptor with a few flipped bits, so medians tie often
    desc = np.zeros((offsets[-1], 32), np.uint8)
    for p, n in enumerate(sizes):
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        for i in range(n):
            flips = np.zeros(256, np.uint8)
            flips[rng.choice(256, rng.integers(0, 12), replace=False)] = 1
            desc[offsets[p] + i] = base ^ np.packbits(flips)
    want = oracle.distinctive_descriptors(desc, offsets)
    got = capi.Matcher().distinctive_descriptors(desc, offsets)
    assert np.array_equal(got, want)
    assert want[12] == -1 and want[0] == 0


@pytest.mark.parametrize("max_dist,ori", [(64, True), (50, False)])
def test_search_by_projection_reloc_and_loop_variants(feats, max_dist, ori):
    """SearchByProjection(Frame, KeyFrame, set, th, ORBdist) (:303-440) and SearchByProjection(KeyFrame, Scw, ...) (:443-550):
    mode 0 with another distance threshold, no stereo check, every assigned feature blocks."""
    grid = (0.0, 0.0, 0.1, 0.1)
    m = capi.Matcher(0.8, ori)
    for a, b, seed in [(0, 1, 21), (2, 1, 22)]:
        q, qd, qa, kb, db, ur, occ = _proj_scene(feats, a, b, 0, "neither", seed)
        q["flags"] |= 2
        q["er_max"] = np.float32(np.inf)
        want_n, want = oracle.search_by_projection(0, q, qd, qa, kb, db, None, occ, grid, 0.8, ori, max_dist)
        got_n, got = m.search_by_projection(0, q, qd, qa, kb, db, None, occ, grid, max_dist)
        assert want_n > 50
        assert got_n == want_n and np.array_equal(got, want)


@pytest.mark.parametrize("chi2,max_dist", [(True, 50), (False, 50), (False, 100)])
def test_search_by_projection_best_fuse_and_sim3(feats, chi2, max_dist):
    grid = (0.0, 0.0, 0.1, 0.1)
    m = capi.Matcher()
    inv_sigma2 = (np.float32(1) / (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2).astype(np.float32)
    for a, b, seed in [(0, 1, 31), (1, 2, 32), (2, 0, 33)]:
        q, qd, qa, kb, db, ur, occ = _proj_scene(feats, a, b, 1, "map", seed)
        ur[::7] = 0.0                                           # mvuRight == 0 counts as stereo for the chi2 gate (>= 0)
        want_i, want_d = oracle.search_by_projection_best(q, qd, kb, db, ur, grid, max_dist, chi2, inv_sigma2)
        got_i, got_d = m.search_by_projection_best(q, qd, kb, db, ur, grid, max_dist, chi2, inv_sigma2)
        assert (want_i >= 0).sum() > 100
        assert np.array_equal(got_i, want_i) and np.array_equal(got_d, want_d)
    # monocular keyframe (no mvuRight)
    q, qd, qa, kb, db, ur, occ = _proj_scene(feats, 0, 2, 1, "map", 34)
    want_i, want_d = oracle.search_by_projection_best(q, qd, kb, db, None, grid, max_dist, chi2, inv_sigma2)
    got_i, got_d = m.search_by_projection_best(q, qd, kb, db, None, grid, max_dist, chi2, inv_sigma2)
    assert np.array_equal(got_i, want_i) and np.array_equal(got_d, want_d)


@pytest.mark.parametrize("only_stereo,ori,mono", [(False, True, False), (True, True, False), (False, False, True)])
def test_search_for_triangulation(feats, only_stereo, ori, mono):
    rng = np.random.default_rng(41)
    m = capi.Matcher(0.6, ori)
    sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    sig2 = (sf * sf).astype(np.float32)
    for a, b in [(0, 1), (1, 2), (3, 0)]:
        (k1, d1), (k2, d2) = feats[a], feats[b]
        fv1, t1 = _fv(d1)
        fv2, t2 = _fv(d2)
        mp1 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
        mp2 = (rng.random(len(k2)) < 0.3).astype(np.uint8)
        ur1 = None if mono else np.where(rng.random(len(k1)) < 0.5, k1["x"] - 10, -1).astype(np.float32)
        ur2 = None if mono else np.where(rng.random(len(k2)) < 0.5, k2["x"] - 10, -1).astype(np.float32)
        # a fundamental matrix of a small sideways motion: epipolar lines are (nearly) the image rows
        F12 = np.array([[0, 0, 0.0004], [0, 0, -1.0], [-0.0003, 1.0, 0.2]], np.float32)
        ex, ey = np.float32(5000.0), np.float32(240.0)
        wn, wm = oracle.search_for_triangulation(k1, d1, mp1, ur1, fv1, k2, d2, mp2, ur2, fv2, F12, ex, ey, sf, sig2, only_stereo, ori)
        gn, gm = m.search_for_triangulation(k1, d1, mp1, ur1, t1, k2, d2, mp2, ur2, t2, F12, ex, ey, sf, sig2, only_stereo)
        if not only_stereo:
            assert wn > 30
        assert gn == wn and np.array_equal(gm, wm)
    # epipole in the middle of the image: the "too close to the epipole" rule fires for monocular features
    (k1, d1), (k2, d2) = feats[0], feats[1]
    fv1, t1 = _fv(d1)
    fv2, t2 = _fv(d2)
    z1, z2 = np.zeros(len(k1), np.uint8), np.zeros(len(k2), np.uint8)
    F12 = np.array([[0, 0, 0.0004], [0, 0, -1.0], [-0.0003, 1.0, 0.2]], np.float32)
    wn, wm = oracle.search_for_triangulation(k1, d1, z1, None, fv1, k2, d2, z2, None, fv2, F12, 320.0, 240.0, sf * 400, sig2, False, ori)
    gn, gm = m.search_for_triangulation(k1, d1, z1, None, t1, k2, d2, z2, None, t2, F12, 320.0, 240.0, sf * 400, sig2, False)
    assert gn == wn and np.array_equal(gm, wm)


def _rand_feature_sets(rng, na, nb, clustered):
    """Two descriptor sets with planted near-duplicates; `clustered` squeezes them into few vocabulary nodes."""
    def rnd(n):
        d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        if clustered and n:
            proto = rng.integers(0, 256, size=(3, 32), dtype=np.uint8)          # 3 tight clusters -> few, big nodes
            d = proto[rng.integers(0, 3, n)] ^ (rng.integers(0, 256, size=(n, 32), dtype=np.uint8) & rng.integers(0, 256, size=(n, 32), dtype=np.uint8) & 0x11)
        return d
    da = rnd(na)
    db = rnd(nb)
    m = min(na, nb)
    if m:
        idx = rng.permutation(m)[: max(1, m * 2 // 3)]
        flips = np.zeros((len(idx), 32), np.uint8)
        for r in range(len(idx)):
            for b in rng.integers(0, 256, rng.integers(0, 40)):
                flips[r, b >> 3] ^= np.uint8(1 << (b & 7))
        db[idx] = da[idx] ^ flips
    aa = rng.uniform(0, 360, na).astype(np.float32)
    ab = rng.uniform(0, 360, nb).astype(np.float32)
    if m:
        ab[idx] = (aa[idx] + rng.choice([0.0, 0.0, 0.0, 90.0], len(idx)).astype(np.float32) + rng.uniform(-3, 3, len(idx)).astype(np.float32)) % np.float32(360.0)
    return da, aa, db, ab


def test_search_by_bow_randomized_sizes():
    """Both SearchByBoW variants on seeded random feature sets: sizes 0, 1, around the wave width, thousands; few big
    vocabulary nodes (general > 64 path) and many small ones; random valid flags, ratios and orientation checks."""
    rng = np.random.default_rng(77)
    sizes = [0, 1, 2, 5, 63, 64, 65, 130, 500, 1000, 2600]
    checked = 0
    for t in range(36):
        na, nb = int(rng.choice(sizes)), int(rng.choice(sizes))
        clustered = bool(rng.integers(0, 3) == 0) and max(na, nb) <= 1000
        ratio, ori = float(rng.choice([0.6, 0.7, 0.75, 0.9])), bool(rng.integers(0, 2))
        da, aa, db, ab = _rand_feature_sets(rng, na, nb, clustered)
        fva, ta = _fv(da)
        fvb, tb = _fv(db)
        va = (rng.random(na) < 0.7).astype(np.uint8)
        vb = (rng.random(nb) < 0.7).astype(np.uint8)
        m = capi.Matcher(ratio, ori)
        wn, w = oracle.search_by_bow(da, aa, va, fva, db, ab, fvb, ratio, ori)
        gn, g = m.search_by_bow(da, aa, va, ta, db, ab, tb)
        assert gn == wn and np.array_equal(g, w), ("KF-F", t, na, nb, clustered, ratio, ori)
        wn2, w2 = oracle.search_by_bow_kk(da, aa, va, fva, db, ab, vb, fvb, ratio, ori)
        gn2, g2 = m.search_by_bow_kk(da, aa, va, ta, db, ab, vb, tb)
        assert gn2 == wn2 and np.array_equal(g2, w2), ("KF-KF", t, na, nb, clustered, ratio, ori)
        checked += wn + wn2
    assert checked > 500                                           # the planted duplicates do get matched


def test_search_by_bow_features_competing_for_the_same_partner():
    """Many features of a node want the SAME Frame feature (the greedy "already taken" rule; in the kernel: the test that
    tells whether the four rows of a group are coupled, and its row-by-row fallback), nodes of 1..40 features; both
    variants.  tools/stress_parity.py runs thousands of these."""
    rng = np.random.default_rng(4242)
    total = 0
    for t in range(40):
        na, nb = (int(rng.integers(1, 80)), int(rng.integers(1, 80))) if t % 4 == 0 else (int(rng.integers(1, 1500)), int(rng.integers(1, 1500)))
        reuse, maxflip = float(rng.choice([0.05, 0.2, 0.6])), int(rng.choice([2, 8, 30]))
        ratio, ori = float(rng.choice([0.7, 0.9, 1.0])), bool(rng.integers(0, 2))
        db = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
        pool = rng.integers(0, nb, max(1, int(nb * reuse)))
        da = db[pool[rng.integers(0, len(pool), na)]].copy()
        for r in range(na):
            for b in rng.integers(0, 256, rng.integers(0, maxflip + 1)):
                da[r, b >> 3] ^= np.uint8(1 << (b & 7))
        aa, ab = rng.uniform(0, 360, na).astype(np.float32), rng.uniform(0, 360, nb).astype(np.float32)
        va, vb = (rng.random(na) < 0.8).astype(np.uint8), (rng.random(nb) < 0.8).astype(np.uint8)
        fva, ta = _fv(da)
        fvb, tb = _fv(db)
        m = capi.Matcher(ratio, ori)
        wn, w = oracle.search_by_bow(da, aa, va, fva, db, ab, fvb, ratio, ori)
        gn, g = m.search_by_bow(da, aa, va, ta, db, ab, tb)
        assert gn == wn and np.array_equal(g, w), ("KF-F", t, na, nb)
        wn2, w2 = oracle.search_by_bow_kk(da, aa, va, fva, db, ab, vb, fvb, ratio, ori)
        gn2, g2 = m.search_by_bow_kk(da, aa, va, ta, db, ab, vb, tb)
        assert gn2 == wn2 and np.array_equal(g2, w2), ("KF-KF", t, na, nb)
        total += wn + wn2
    assert total > 2000


def test_search_by_bow_rotation_histogram_ties():
    """ComputeThreeMaxima on the device (three wave maxima) against the reference's sequential scan: planted matches whose
    rotation differences fall into a few bins with EQUAL counts (ties go to the lower bin), with one dominant bin (the
    0.1 rule drops the others) and with fewer than three occupied bins."""
    rng = np.random.default_rng(99)
    for t, (rots, per_bin) in enumerate([((0, 36, 72, 108), (20, 20, 20, 20)), ((0, 36, 72, 108, 144), (7, 7, 9, 7, 9)),
                                         ((0, 72), (40, 3)), ((36,), (25,)), ((0, 36, 72), (30, 3, 2)), ((0, 36, 72, 108), (5, 5, 5, 4))]):
        n = sum(per_bin)
        da = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        db = da.copy()
        db[:, 0] ^= 1                                              # distance 1: every planted pair matches
        aa = rng.uniform(100, 200, n).astype(np.float32)
        rot = np.concatenate([np.full(c, r, np.float32) for r, c in zip(rots, per_bin)])
        ab = ((aa - rot - np.float32(3.0)) % np.float32(360.0)).astype(np.float32)     # rot = angA - angB lands mid-bin
        fva, ta = _fv(da)
        fvb, tb = _fv(db)
        va = np.ones(n, np.uint8)
        m = capi.Matcher(0.9, True)
        wn, w = oracle.search_by_bow(da, aa, va, fva, db, ab, fvb, 0.9, True)
        gn, g = m.search_by_bow(da, aa, va, ta, db, ab, tb)
        assert gn == wn and np.array_equal(g, w), t
        wn2, w2 = oracle.search_by_bow_kk(da, aa, va, fva, db, ab, va, fvb, 0.9, True)
        gn2, g2 = m.search_by_bow_kk(da, aa, va, ta, db, ab, va, tb)
        assert gn2 == wn2 and np.array_equal(g2, w2), t
        assert 0 < wn <= n and (len(rots) <= 3 or wn < n)          # something is kept, and a 4th bin is dropped


def _rand_keypoints(rng, n, w=640, h=480, level0_frac=0.5):
    k = np.zeros(n, oracle.KP_DTYPE if hasattr(oracle, "KP_DTYPE") else capi.KP_DTYPE)
    k["x"] = rng.uniform(16, w - 16, n).astype(np.float32)
    k["y"] = rng.uniform(16, h - 16, n).astype(np.float32)
    k["octave"] = np.where(rng.random(n) < level0_frac, 0, rng.integers(1, 8, n)).astype(np.int32)
    k["size"] = 31.0
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    k["response"] = rng.integers(8, 200, n).astype(np.float32)
    k["class_id"] = -1
    return k


def test_search_for_initialization_randomized_sizes():
    """SearchForInitialization on seeded random keypoint sets: 0 / 1 / wave-width / thousands of features, dense
    clusters (many candidates per window, match stealing), windows from 5 to 300 px, shifted grids."""
    rng = np.random.default_rng(99)
    sizes = [0, 1, 3, 63, 64, 65, 200, 900, 2100]
    total = 0
    for t in range(24):
        n1, n2 = int(rng.choice(sizes)), int(rng.choice(sizes))
        window = int(rng.choice([5, 20, 100, 300]))
        ratio, ori = float(rng.choice([0.7, 0.9])), bool(rng.integers(0, 2))
        d1, a1, d2, a2 = _rand_feature_sets(rng, n1, n2, False)
        k1, k2 = _rand_keypoints(rng, n1), _rand_keypoints(rng, n2)
        k1["angle"], k2["angle"] = a1, a2
        m12 = min(n1, n2)
        if m12:                                                    # planted duplicates sit near each other
            k2["x"][:m12] = np.clip(k1["x"][:m12] + rng.uniform(-window, window, m12).astype(np.float32) * 0.7, 0, 639)
            k2["y"][:m12] = np.clip(k1["y"][:m12] + rng.uniform(-window, window, m12).astype(np.float32) * 0.7, 0, 479)
            k2["octave"][:m12] = 0
            k1["octave"][:m12] = 0
        if t % 5 == 4 and n2 > 10:                                 # one crowded spot: dozens of candidates per window
            k2["x"][: n2 // 2] = np.float32(320) + rng.uniform(-8, 8, n2 // 2).astype(np.float32)
            k2["y"][: n2 // 2] = np.float32(240) + rng.uniform(-8, 8, n2 // 2).astype(np.float32)
        grid = GRID_640 if t % 3 else (-9.5, -4.25, 64.0 / 660.0, 48.0 / 490.0)
        prev_ref = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1), dtype=np.float32).reshape(n1, 2)
        prev_gpu = prev_ref.copy()
        m = capi.Matcher(ratio, ori)
        wn, w = oracle.search_for_init(k1, d1, k2, d2, grid, prev_ref, window, ratio, ori)
        gn, g = m.search_for_initialization(k1, d1, k2, d2, grid, prev_gpu, window)
        assert gn == wn and np.array_equal(g, w) and prev_gpu.tobytes() == prev_ref.tobytes(), (t, n1, n2, window, ratio, ori)
        total += wn
    assert total > 200


def test_projection_family_and_triangulation_randomized_subsets(feats):
    """The SearchByProjection family, its best-candidate form and SearchForTriangulation on random SUBSETS of the
    scene features: query / feature counts of 0, 1, 63..65 and ~1000, all modes, with and without stereo."""
    rng = np.random.default_rng(123)
    grid = (0.0, 0.0, 0.1, 0.1)
    sizes = [0, 1, 2, 63, 64, 65, 300, 1000]
    inv_sigma2 = (np.float32(1) / (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2).astype(np.float32)
    sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    sig2 = (sf * sf).astype(np.float32)
    total = 0
    for t in range(20):
        a, b = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        mode = int(rng.integers(0, 2))
        level_mode = ["neither", "forward", "backward"][int(rng.integers(0, 3))] if mode == 0 else "map"
        q, qd, qa, kb, db, ur, occ = _proj_scene(feats, a, b, mode, level_mode, 500 + t)
        nq, n = min(int(rng.choice(sizes)), len(q)), min(int(rng.choice(sizes)), len(kb))
        qs = np.sort(rng.permutation(len(q))[:nq])
        fs = np.sort(rng.permutation(len(kb))[:n])
        q, qd, qa = q[qs].copy(), np.ascontiguousarray(qd[qs]), np.ascontiguousarray(qa[qs])
        kb, db, ur, occ = kb[fs].copy(), np.ascontiguousarray(db[fs]), np.ascontiguousarray(ur[fs]), np.ascontiguousarray(occ[fs])
        ori = bool(rng.integers(0, 2))
        m = capi.Matcher(0.8, ori)
        wn, w = oracle.search_by_projection(mode, q, qd, qa, kb, db, ur, occ, grid, 0.8, ori)
        gn, g = m.search_by_projection(mode, q, qd, qa, kb, db, ur, occ, grid)
        assert gn == wn and np.array_equal(g, w), ("projection", t, mode, level_mode, nq, n, ori)
        chi2 = bool(rng.integers(0, 2))
        wi, wd = oracle.search_by_projection_best(q, qd, kb, db, ur, grid, 64, chi2, inv_sigma2)
        gi, gd = m.search_by_projection_best(q, qd, kb, db, ur, grid, 64, chi2, inv_sigma2)
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd), ("best", t, nq, n, chi2)
        total += wn + int((wi >= 0).sum())
        # triangulation between the two subsets (frame a features restricted to the query subset)
        (k1, d1) = feats[a]
        k1, d1 = k1[qs].copy(), np.ascontiguousarray(d1[qs])
        fv1, t1 = _fv(d1)
        fv2, t2 = _fv(db)
        mp1 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
        mp2 = (rng.random(len(kb)) < 0.3).astype(np.uint8)
        F12 = np.array([[0, 0, 0.0004], [0, 0, -1.0], [-0.0003, 1.0, 0.2]], np.float32)
        mt = capi.Matcher(0.6, ori)
        twn, twm = oracle.search_for_triangulation(k1, d1, mp1, None, fv1, kb, db, mp2, None, fv2, F12, 5000.0, 240.0, sf, sig2, False, ori)
        tgn, tgm = mt.search_for_triangulation(k1, d1, mp1, None, t1, kb, db, mp2, None, t2, F12, 5000.0, 240.0, sf, sig2, False)
        assert tgn == twn and np.array_equal(tgm, twm), ("triangulation", t, nq, n, ori)
        total += twn
    assert total > 100


def test_c5_full_size_1000_keyframe_db():
    """BASELINE configs[4] at full size: a 752x480 stream frame against a 1000-keyframe descriptor DB in HBM (the
    Relocalization candidate loop, reference src/Tracking.cc:1471-1492), DBoW2-shaped vocabulary (k=10, L=5 here).
    Size-independent properties over all 1000 pairs (the query recognises the keyframes of its own scene; the batch
    equals itself when repeated and when the matcher builds the feature vectors itself instead of reading the
    store's CSR; match arrays are injective) + oracle parity on a seeded sample of 40 pairs."""
    import torch
    W, H, n_kf, nq = 752, 480, 1000, 3
    dev = torch.device("cuda", 0)
    ex, mt = capi.Extractor(), capi.Matcher(0.75, True)
    cap = ex.max_keypoints
    F = n_kf + nq
    def z(n, dt):
        # torch fills on ITS stream; the library's streams are non-blocking and nothing orders them behind it: a fill still in
        # flight would overwrite what the library writes (seen once as rows of zeros in a result) -- wait for it here
        t = torch.zeros(n, dtype=dt, device=dev)
        torch.cuda.synchronize()
        return t
    d_kps, d_desc, d_counts, d_node = z(F * cap * 28, torch.uint8), z(F * cap * 32, torch.uint8), z(F, torch.int32), z(F * cap, torch.int16)
    valid_np = np.stack([synth.synth_valid_flags(cap, 7000 + i) for i in range(F)])
    d_valid = torch.from_numpy(valid_np).to(dev)
    tree = synth.synth_vocab_tree_balanced(10, 5, seed=5)
    voc = capi.Vocabulary(tree)
    nn = voc.level_nodes(3)                                          # level 2: 100 nodes
    q_scene = [17, 60, 101]
    q_frames = np.stack([synth.synth_sequence(8 * s + 3, 1, W, H, noise=5)[0] for s in q_scene])
    for k0 in range(0, F, 40):
        n = min(40, F - k0)
        fr = synth.synth_sequence(k0, n, W, H) if k0 + n <= n_kf else np.concatenate([synth.synth_sequence(k0, n_kf - k0, W, H), q_frames])[:n]
        d_b = torch.from_numpy(fr).to(dev)
        ex.extract_batch_device(d_b.data_ptr(), n, H, W, W, W * H, d_kps.data_ptr() + k0 * cap * 28, d_desc.data_ptr() + k0 * cap * 32,
                                cap, d_counts.data_ptr() + k0 * 4)
        ex.sync()
    voc.transform_device(mt, d_desc.data_ptr(), d_counts.data_ptr(), F, cap, 3, d_node_of=d_node.data_ptr())
    d_ck, d_cs, d_cc = z(F * cap, torch.int32), z(F * nn, torch.int16), z(F * nn, torch.int16)
    d_cd = z(F * cap * 32, torch.uint8)
    mt.build_csr_desc_device(d_node.data_ptr(), d_counts.data_ptr(), d_desc.data_ptr(), F, cap, nn, d_ck.data_ptr(), d_cs.data_ptr(),
                             d_cc.data_ptr(), d_cd.data_ptr())
    base = dict(desc=d_desc.data_ptr(), kps=d_kps.data_ptr(), valid=d_valid.data_ptr(), counts=d_counts.data_ptr(),
                node_of=d_node.data_ptr(), cap=cap, n_frames=F, n_nodes=nn)
    with_csr = dict(base, csr_keys=d_ck.data_ptr(), csr_start=d_cs.data_ptr(), csr_cnt=d_cc.data_ptr(), csr_desc=d_cd.data_ptr())
    kf_idx = torch.arange(n_kf, dtype=torch.int32, device=dev)
    # the query form (orb_match_bow_query_device): all three queries against the 1000 keyframes in ONE call
    q_idx = torch.arange(n_kf, n_kf + nq, dtype=torch.int32, device=dev)
    d_mq, d_nq = z(nq * n_kf * cap, torch.int32), z(nq * n_kf, torch.int32)
    mt.match_bow_query_device(with_csr, kf_idx.data_ptr(), n_kf, q_idx.data_ptr(), nq, d_mq.data_ptr(), d_nq.data_ptr())
    mt.sync()
    mq, nmq = d_mq.cpu().numpy().reshape(nq, n_kf, cap), d_nq.cpu().numpy().reshape(nq, n_kf)
    counts = d_counts.cpu().numpy()
    kps = d_kps.cpu().numpy().view(capi.KP_DTYPE).reshape(F, cap)
    desc = d_desc.cpu().numpy().reshape(F, cap, 32)
    rng = np.random.default_rng(11)
    for q in range(nq):
        f_idx = torch.full((n_kf,), n_kf + q, dtype=torch.int32, device=dev)
        res = []
        for store in (with_csr, with_csr, base):
            d_m, d_n = z(n_kf * cap, torch.int32), z(n_kf, torch.int32)
            mt.match_bow_batch_device(store, kf_idx.data_ptr(), f_idx.data_ptr(), n_kf, d_m.data_ptr(), d_n.data_ptr())
            mt.sync()
            res.append((d_m.cpu().numpy().reshape(n_kf, cap), d_n.cpu().numpy()))
        (m0, n0), (m1, n1), (m2, n2) = res
        assert np.array_equal(m0, m1) and np.array_equal(n0, n1)            # idempotent
        assert np.array_equal(m0, m2) and np.array_equal(n0, n2)            # store CSR == CSR built inside the matcher
        assert np.array_equal(n0, nmq[q]) and np.array_equal(m0[:, :int(counts[n_kf + q])], mq[q][:, :int(counts[n_kf + q])])   # query form == pair kernel, all 1000 keyframes
        nqf = int(counts[n_kf + q])
        assert np.all(n0 == (m0[:, :nqf] >= 0).sum(axis=1))                  # rows are written for the query's features only
        for kf in rng.choice(n_kf, 25, replace=False):                      # a keyframe feature is matched at most once
            mm = m0[kf, :nqf][m0[kf, :nqf] >= 0]
            assert mm.size == np.unique(mm).size and (mm.size == 0 or mm.max() < counts[kf])
        own = np.arange(8 * q_scene[q], 8 * q_scene[q] + 8)
        assert n0[own].min() > 4 * np.delete(n0, own).max() / 3 and n0.argmax() in own   # recognises its own scene
        kq, dq = kps[n_kf + q, :nqf], desc[n_kf + q, :nqf]
        fvq = oracle.featvec_from_nodes(oracle.vocab_transform(tree, dq, 3)[1])
        sample = list(rng.choice(n_kf, 10, replace=False)) + list(own[:4])
        for kf in sample:
            n = int(counts[kf])
            fvk = oracle.featvec_from_nodes(oracle.vocab_transform(tree, desc[kf, :n], 3)[1])
            wn, wm = oracle.search_by_bow(desc[kf, :n], kps[kf, :n]["angle"], valid_np[kf][:n], fvk, dq, kq["angle"], fvq, 0.75, True)
            assert wn == n0[kf] and np.array_equal(wm, m0[kf, :nqf]), (q, kf)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0, 0]])
def test_keyframe_db_sharded_by_keyframe(devices):
    """orb_multi_db_* / orb_multi_match_bow_batch (BASELINE configs[4]): the keyframe DB cut into contiguous
    blocks of keyframes, one shard (matcher handle, store slice, host thread) per listed device, the query replicated, results
    written in place.  On the one-GPU box the device is listed several times (partition, threads, merge; no second GPU is
    exercised).  Every keyframe's result equals the single-store batch matcher's and, on a sample, the oracle's."""
    import torch
    W, H, n_kf = 400, 300, 37
    dev = torch.device("cuda", 0)
    ex, mt = capi.Extractor(600), capi.Matcher(0.75, True)
    cap = ex.max_keypoints
    F = n_kf + 2
    def z(n, dt):
        # torch fills on ITS stream; the library's streams are non-blocking and nothing orders them behind it: a fill still in
        # flight would overwrite what the library writes (seen once as rows of zeros in a result) -- wait for it here
        t = torch.zeros(n, dtype=dt, device=dev)
        torch.cuda.synchronize()
        return t
    d_kps, d_desc, d_counts, d_node = z(F * cap * 28, torch.uint8), z(F * cap * 32, torch.uint8), z(F, torch.int32), z(F * cap, torch.int16)
    frames = np.concatenate([synth.synth_sequence(0, n_kf, W, H), synth.synth_sequence(11, 1, W, H, noise=5), synth.synth_sequence(26, 1, W, H, noise=4)])
    d_b = torch.from_numpy(frames).to(dev)
    ex.extract_batch_device(d_b.data_ptr(), F, H, W, W, W * H, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_counts.data_ptr())
    ex.sync()
    tree = synth.synth_vocab_tree_balanced(10, 4, seed=9)
    voc = capi.Vocabulary(tree)
    nn = voc.level_nodes(2)
    voc.transform_device(mt, d_desc.data_ptr(), d_counts.data_ptr(), F, cap, 2, d_node_of=d_node.data_ptr())
    mt.sync()
    counts = d_counts.cpu().numpy()
    kps = d_kps.cpu().numpy().view(capi.KP_DTYPE).reshape(F, cap)
    desc = d_desc.cpu().numpy().reshape(F, cap, 32)
    node = d_node.cpu().numpy().view(np.uint16).reshape(F, cap)
    valid = np.stack([synth.synth_valid_flags(cap, 900 + i) for i in range(n_kf)])
    db = capi.MultiKeyframeDB(devices, desc[:n_kf], kps[:n_kf], valid, counts[:n_kf], node[:n_kf], nn)
    assert db.shards == len(devices)
    rng = np.random.default_rng(3)
    for q in (n_kf, n_kf + 1):
        nq = int(counts[q])
        match, nm = db.match(desc[q, :nq], kps[q, :nq], node[q, :nq], 0.75, True)
        assert nm.min() >= 0 and nm.max() > 60                               # the query's own scene is in the DB
        for kf in list(rng.choice(n_kf, 8, replace=False)) + [int(nm.argmax())]:
            n = int(counts[kf])
            fvk = oracle.featvec_from_nodes(oracle.vocab_transform(tree, desc[kf, :n], 2)[1])
            fvq = oracle.featvec_from_nodes(oracle.vocab_transform(tree, desc[q, :nq], 2)[1])
            wn, wm = oracle.search_by_bow(desc[kf, :n], kps[kf, :n]["angle"], valid[kf][:n], fvk, desc[q, :nq], kps[q, :nq]["angle"], fvq, 0.75, True)
            assert wn == nm[kf] and np.array_equal(wm, match[kf, :nq]), (devices, q, kf)
        assert np.all(nm == (match[:, :nq] >= 0).sum(axis=1))
    # valid = NULL means every keyframe feature has a good MapPoint; an empty query gives no matches
    db2 = capi.MultiKeyframeDB(devices, desc[:n_kf], kps[:n_kf], None, counts[:n_kf], node[:n_kf], nn)
    m2, n2 = db2.match(desc[n_kf, :int(counts[n_kf])], kps[n_kf, :int(counts[n_kf])], node[n_kf, :int(counts[n_kf])], 0.75, True)
    assert n2.sum() >= db.match(desc[n_kf, :int(counts[n_kf])], kps[n_kf, :int(counts[n_kf])], node[n_kf, :int(counts[n_kf])], 0.75, True)[1].sum()
    m3, n3 = db2.match(desc[n_kf, :0], kps[n_kf, :0], node[n_kf, :0])
    assert np.all(n3 == 0)
    db.close(); db2.close()


def test_search_by_bow_frames_of_6000_features():
    """Frames as large as the 2 x nFeatures initialisation extractor produces (reference src/Tracking.cc:121): more than
    64 KB of LDS per pair in k_match_bow (the CU's whole 160 KB is available to one workgroup)."""
    ref = oracle.Extractor(6000)
    base = synth.synth_frame(100, 1241, 376, noise=0).astype(np.int16)
    fe = []
    for s in range(2):
        nz = (synth.splitmix64(555 + s, 0, base.size) % np.uint64(13)).astype(np.int16).reshape(base.shape) - 6
        fe.append(ref.extract(np.clip(base + nz, 0, 255).astype(np.uint8)))
    (ka, da), (kb, db) = fe
    assert len(ka) > 4500 and len(kb) > 4500
    fva, ta = _fv(da)
    fvb, tb = _fv(db)
    valid = synth.synth_valid_flags(len(ka), 3)
    m = capi.Matcher(0.7, True)
    wn, wm = oracle.search_by_bow(da, ka["angle"], valid, fva, db, kb["angle"], fvb, 0.7, True)
    n, out = m.search_by_bow(da, ka["angle"], valid, ta, db, kb["angle"], tb)
    assert n == wn and np.array_equal(out, wm) and n > 500
