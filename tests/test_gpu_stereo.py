"""GPU parity of the stereo search (Frame::ComputeStereoMatches, config 3: KITTI-sized pair, 2000 features)."""
import numpy as np
import pytest

import oracle
from orbhip import capi, synth

pytestmark = pytest.mark.gpu

MBF = 386.1448                      # KITTI-00-like constants
MB = MBF / 718.856


def _pair(idx, w, h, nf, natural=False):
    left = synth.synth_natural(idx, w, h) if natural else synth.synth_frame(idx, w, h)
    right = synth.synth_natural_stereo_right(idx, w, h) if natural else synth.synth_stereo_right(idx, w, h)
    exl, exr = capi.Extractor(nf), capi.Extractor(nf)
    kl, dl = exl.extract(left)
    kr, dr = exr.extract(right)
    rl, rr = oracle.Extractor(nf), oracle.Extractor(nf)
    okl, odl = rl.extract(left)
    okr, odr = rr.extract(right)
    assert kl.tobytes() == okl.tobytes() and kr.tobytes() == okr.tobytes()
    assert np.array_equal(dl, odl) and np.array_equal(dr, odr)
    return (exl, exr, kl, dl, kr, dr), (rl, rr)


@pytest.mark.parametrize("idx,w,h,nf", [(100, 1241, 376, 2000), (101, 1241, 376, 2000), (7, 640, 480, 1000)])
def test_stereo_matches_bit_exact(idx, w, h, nf):
    (exl, exr, kl, dl, kr, dr), (rl, rr) = _pair(idx, w, h, nf)
    want_u, want_z = oracle.stereo_matches(rl, rr, kl, dl, kr, dr, MB, MBF)
    got_u, got_z = capi.stereo_match(exl, exr, kl, dl, kr, dr, MB, MBF)
    assert (want_u >= 0).sum() > 0.3 * len(kl)
    assert got_u.tobytes() == want_u.tobytes()
    assert got_z.tobytes() == want_z.tobytes()
    # the synthetic pair has disparity 12 + 8*floor(y/94): the recovered disparities cluster there
    ok = got_u >= 0
    disp = kl["x"][ok] - got_u[ok]
    band = 12 + 8 * np.floor(kl["y"][ok] / 94.0)
    assert np.mean(np.abs(disp - band) < 1.5) > 0.8


def test_stereo_on_natural_statistics_pair():
    """KITTI-sized pair with natural image statistics (synth.synth_natural): extraction of both views and the stereo search
    equal the oracle's bit for bit, and the recovered disparities sit on the synthetic disparity bands."""
    (exl, exr, kl, dl, kr, dr), (rl, rr) = _pair(2, 1241, 376, 2000, natural=True)
    want_u, want_z = oracle.stereo_matches(rl, rr, kl, dl, kr, dr, MB, MBF)
    got_u, got_z = capi.stereo_match(exl, exr, kl, dl, kr, dr, MB, MBF)
    assert got_u.tobytes() == want_u.tobytes() and got_z.tobytes() == want_z.tobytes()
    ok = got_u >= 0
    assert ok.sum() > 0.2 * len(kl)
    disp = kl["x"][ok] - got_u[ok]
    band = 12 + 8 * np.floor(kl["y"][ok] / 94.0)
    assert np.mean(np.abs(disp - band) < 1.5) > 0.7


def test_stereo_no_right_features_and_swapped_baseline():
    (exl, exr, kl, dl, kr, dr), (rl, rr) = _pair(9, 640, 480, 500)
    u, z = capi.stereo_match(exl, exr, kl, dl, kr[:0], dr[:0], MB, MBF)
    assert np.all(u == -1) and np.all(z == -1)
    # tiny maxD (= mbf/mb) leaves (almost) no candidate in range: still identical to the oracle
    wu, wz = oracle.stereo_matches(rl, rr, kl, dl, kr, dr, 10.0, 40.0)
    gu, gz = capi.stereo_match(exl, exr, kl, dl, kr, dr, 10.0, 40.0)
    assert gu.tobytes() == wu.tobytes() and gz.tobytes() == wz.tobytes()


@pytest.mark.parametrize("idx,w,h,nf", [(300, 240, 180, 300), (301, 752, 480, 1000), (302, 1241, 376, 500), (303, 400, 960, 700),
                                        (304, 321, 245, 64), (305, 1000, 300, 3000)])
def test_stereo_other_sizes_and_feature_counts(idx, w, h, nf):
    (exl, exr, kl, dl, kr, dr), (rl, rr) = _pair(idx, w, h, nf)
    for mb, mbf in ((MB, MBF), (0.12, 40.0)):
        want_u, want_z = oracle.stereo_matches(rl, rr, kl, dl, kr, dr, mb, mbf)
        got_u, got_z = capi.stereo_match(exl, exr, kl, dl, kr, dr, mb, mbf)
        assert got_u.tobytes() == want_u.tobytes() and got_z.tobytes() == want_z.tobytes()
    # subsets with awkward counts (the right side scanned 64 at a time)
    rng = np.random.default_rng(idx)
    for nl, nr in ((1, 65), (64, 1), (65, 63), (len(kl), 64)):
        sl = np.sort(rng.permutation(len(kl))[:min(nl, len(kl))])
        sr = np.sort(rng.permutation(len(kr))[:min(nr, len(kr))])
        a = (kl[sl].copy(), np.ascontiguousarray(dl[sl]), kr[sr].copy(), np.ascontiguousarray(dr[sr]))
        want_u, want_z = oracle.stereo_matches(rl, rr, *a, MB, MBF)
        got_u, got_z = capi.stereo_match(exl, exr, *a, MB, MBF)
        assert got_u.tobytes() == want_u.tobytes() and got_z.tobytes() == want_z.tobytes(), (nl, nr)


def test_stereo_right_keypoints_in_any_order_and_more_right_than_left():
    """The coarse search scans only the neighbouring pyramid levels when the right keypoints come level after level (the
    extractor's order); a caller's own order (shuffled here: "first minimum wins" then follows THAT order) and a right
    side larger than the left one go through the full scan -- both identical to the oracle on the same arrays."""
    (exl, exr, kl, dl, kr, dr), (rl, rr) = _pair(310, 752, 480, 1200)
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(kr))
    krs, drs = kr[perm].copy(), np.ascontiguousarray(dr[perm])
    want_u, want_z = oracle.stereo_matches(rl, rr, kl, dl, krs, drs, MB, MBF)
    got_u, got_z = capi.stereo_match(exl, exr, kl, dl, krs, drs, MB, MBF)
    assert (want_u >= 0).sum() > 0.2 * len(kl)
    assert got_u.tobytes() == want_u.tobytes() and got_z.tobytes() == want_z.tobytes()
    # 100 left keypoints (a slice keeps them level-ordered) against all right ones, ordered and shuffled
    sl = np.sort(rng.permutation(len(kl))[:100])
    for k2, d2 in ((kr, dr), (krs, drs)):
        a = (kl[sl].copy(), np.ascontiguousarray(dl[sl]), k2, d2)
        want_u, want_z = oracle.stereo_matches(rl, rr, *a, MB, MBF)
        got_u, got_z = capi.stereo_match(exl, exr, *a, MB, MBF)
        assert got_u.tobytes() == want_u.tobytes() and got_z.tobytes() == want_z.tobytes()


def test_stereo_batch_of_pairs_in_one_launch():
    """orb_stereo_match_batch_device: 5 KITTI-sized pairs extracted as two device batches, the stereo search of all of
    them in ONE launch with the keypoint counts read on the device; every pair bit-exact against the oracle (which
    extracts and searches the pairs one at a time), incl. a pair whose right image is blank (no right features)."""
    import torch
    W, H, nf, S = 1241, 376, 2000, 5
    dev = torch.device("cuda", 0)
    lefts = np.stack([synth.synth_frame(400 + i, W, H) for i in range(S)])
    rights = np.stack([synth.synth_stereo_right(400 + i, W, H) for i in range(S)])
    rights[3] = 77                                               # flat right image: zero right keypoints for pair 3
    exl, exr = capi.Extractor(nf), capi.Extractor(nf)
    cap = exl.max_keypoints
    z = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)
    d_l, d_r = torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev)
    kl, kr = z(S * cap * 28, torch.uint8), z(S * cap * 28, torch.uint8)
    dl, dr = z(S * cap * 32, torch.uint8), z(S * cap * 32, torch.uint8)
    cl, cr = z(S, torch.int32), z(S, torch.int32)
    ur, dp = z(S * cap, torch.float32), z(S * cap, torch.float32)
    torch.cuda.synchronize()
    exl.extract_batch_device(d_l.data_ptr(), S, H, W, W, W * H, kl.data_ptr(), dl.data_ptr(), cap, cl.data_ptr())
    exr.extract_batch_device(d_r.data_ptr(), S, H, W, W, W * H, kr.data_ptr(), dr.data_ptr(), cap, cr.data_ptr())
    capi.stereo_match_batch_device(exl, exr, 0, 0, S, kl.data_ptr(), dl.data_ptr(), cl.data_ptr(), kr.data_ptr(), dr.data_ptr(),
                                   cr.data_ptr(), cap, MB, MBF, ur.data_ptr(), dp.data_ptr())
    exl.sync(); exr.sync(); torch.cuda.synchronize()
    u, zz = ur.cpu().numpy().reshape(S, cap), dp.cpu().numpy().reshape(S, cap)
    nl, nr = cl.cpu().numpy(), cr.cpu().numpy()
    assert nr[3] == 0 and nl[3] > 1000
    for i in range(S):
        rl, rr = oracle.Extractor(nf), oracle.Extractor(nf)
        k1, d1 = rl.extract(lefts[i])
        k2, d2 = rr.extract(rights[i])
        assert len(k1) == nl[i] and len(k2) == nr[i]
        wu, wz = oracle.stereo_matches(rl, rr, k1, d1, k2, d2, MB, MBF)
        assert u[i, :nl[i]].tobytes() == wu.tobytes() and zz[i, :nl[i]].tobytes() == wz.tobytes(), i
    assert np.all(u[3, :nl[3]] == -1)


def test_pipelined_stereo_steps_without_a_sync_in_between():
    """Six steps of (extract left batch, extract right batch, stereo search) issued back to back, the extractors' output
    buffers REUSED by every step and only the (u_right, depth) results kept per step: the search of step k reads the right
    handle's pyramid and the right keypoints on the left handle's stream, so the right extraction of step k + 1 must be
    ordered behind it (orb_stereo_match_batch_device adds that edge; without it the two raced).  Every step bit-exact."""
    import torch
    W, H, nf, S, STEPS = 752, 480, 1200, 3, 6
    dev = torch.device("cuda", 0)
    exl, exr = capi.Extractor(nf), capi.Extractor(nf)
    cap = exl.max_keypoints
    z = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)
    lefts = [np.stack([synth.synth_frame(600 + 10 * k + i, W, H) for i in range(S)]) for k in range(STEPS)]
    rights = [np.stack([synth.synth_stereo_right(600 + 10 * k + i, W, H) for i in range(S)]) for k in range(STEPS)]
    d_l = [torch.from_numpy(a).to(dev) for a in lefts]
    d_r = [torch.from_numpy(a).to(dev) for a in rights]
    kl, kr = z(S * cap * 28, torch.uint8), z(S * cap * 28, torch.uint8)
    dl, dr = z(S * cap * 32, torch.uint8), z(S * cap * 32, torch.uint8)
    cl, cr = z(S, torch.int32), z(S, torch.int32)
    ur = [z(S * cap, torch.float32) for _ in range(STEPS)]
    dp = [z(S * cap, torch.float32) for _ in range(STEPS)]
    torch.cuda.synchronize()
    for k in range(STEPS):
        exl.extract_batch_device(d_l[k].data_ptr(), S, H, W, W, W * H, kl.data_ptr(), dl.data_ptr(), cap, cl.data_ptr())
        exr.extract_batch_device(d_r[k].data_ptr(), S, H, W, W, W * H, kr.data_ptr(), dr.data_ptr(), cap, cr.data_ptr())
        capi.stereo_match_batch_device(exl, exr, 0, 0, S, kl.data_ptr(), dl.data_ptr(), cl.data_ptr(), kr.data_ptr(), dr.data_ptr(),
                                       cr.data_ptr(), cap, MB, MBF, ur[k].data_ptr(), dp[k].data_ptr())
    exl.sync(); exr.sync(); torch.cuda.synchronize()
    for k in range(STEPS):
        u, zz = ur[k].cpu().numpy().reshape(S, cap), dp[k].cpu().numpy().reshape(S, cap)
        for i in range(S):
            rl, rr = oracle.Extractor(nf), oracle.Extractor(nf)
            k1, d1 = rl.extract(lefts[k][i])
            k2, d2 = rr.extract(rights[k][i])
            wu, wz = oracle.stereo_matches(rl, rr, k1, d1, k2, d2, MB, MBF)
            assert u[i, :len(k1)].tobytes() == wu.tobytes() and zz[i, :len(k1)].tobytes() == wz.tobytes(), (k, i)
    exl.close(); exr.close()


def test_stereo_ignores_right_keypoints_that_lie_in_no_image_row():
    """Right keypoints whose y is NaN, infinite or far outside the image belong to no row of the per-row table
    (k_stereo_rows): the reference would index vRowIndices out of range with them; here they are simply never candidates --
    the result equals the oracle's on the arrays without them, and nothing is written out of bounds."""
    (exl, exr, kl, dl, kr, dr), (rl, rr) = _pair(311, 752, 480, 1200)
    want_u, want_z = oracle.stereo_matches(rl, rr, kl, dl, kr, dr, MB, MBF)
    rng = np.random.default_rng(9)
    bad = kr[:64].copy()
    bad["y"] = np.resize(np.array([np.nan, np.inf, -np.inf, 1e9, -1e9, 3e38, -3e38, 479.9 + 1000], np.float32), 64)
    bad["octave"] = np.resize(np.array([0, 7, 3, 100, -5], np.int32), 64)
    krb = np.concatenate([kr, bad])
    drb = np.concatenate([dr, rng.integers(0, 256, (64, 32), dtype=np.uint8)])
    got_u, got_z = capi.stereo_match(exl, exr, kl, dl, krb, np.ascontiguousarray(drb), MB, MBF)
    assert got_u.tobytes() == want_u.tobytes() and got_z.tobytes() == want_z.tobytes()
