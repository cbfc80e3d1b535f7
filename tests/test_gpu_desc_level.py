"""GPU parity tests of the level-resident descriptor stage (csrc/orb_desc_level.hip: one staging + one 7x7 blur per pyramid
level region, then all of the region's keypoints -- the structure of reference src/ORBextractor.cc:1118-1136) against the
oracle and against the per-keypoint kernel (csrc/orb_desc.hip), through the C ABI.  ORB_DESC_LEVEL=1 switches the path on (it is
bit-exact but not faster, so it is off by default), ORB_DESC_LEVEL_MIN_FRAMES=1 makes launches of any size take it for the levels
the handle's plan covers."""
import numpy as np
import pytest

import oracle
from orbhip import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _level_path_for_every_launch(monkeypatch):
    monkeypatch.setenv("ORB_DESC_LEVEL", "1")                       # (off by default: DESIGN.md 3.2 has the measurements)
    monkeypatch.setenv("ORB_DESC_LEVEL_MIN_FRAMES", "1")


def _check(imgs, nfeatures=1000, levels=8, ini=20, mn=7, sf=1.2, gauss=None, min_regions=1):
    ex = capi.Extractor(nfeatures, sf, levels, ini, mn)
    ref = oracle.Extractor(nfeatures, sf, levels, ini, mn)
    if gauss is not None:
        ref.set_gaussian(ex.set_gaussian(gauss))
    got = ex.extract_batch(np.stack(imgs)) if len(imgs) > 1 else [ex.extract(imgs[0])]
    first, nreg = ex.desc_plan()
    assert nreg >= min_regions and (first < levels) == (nreg > 0), (first, nreg)
    total = 0
    for i, im in enumerate(imgs):
        rk, rd = ref.extract(im)
        k, d = got[i]
        assert len(k) == len(rk), (i, len(k), len(rk))
        for name in ("octave", "x", "y", "response", "size", "angle", "class_id"):
            assert np.array_equal(k[name], rk[name]), "frame %d keypoint field %s" % (i, name)
        assert k.tobytes() == rk.tobytes()
        bad = np.nonzero(np.any(d != rd, axis=1))[0]
        assert bad.size == 0, "frame %d: %d descriptors differ, first at %d (octave %d, x %g y %g)" % (
            i, bad.size, bad[0], rk["octave"][bad[0]], rk["x"][bad[0]], rk["y"][bad[0]])
        total += int(np.sum(rk["octave"] >= first))
    ex.close()
    return first, nreg, total


def test_vga_plan_covers_the_upper_levels():
    """640x480 / 1000 features / 8 levels: levels >= 3 fit a workgroup's share of the LDS whole, level 2 as row tiles."""
    first, nreg, n = _check([synth.synth_frame(0)])
    assert first <= 3 and nreg >= 8 - first
    assert n > 300                                                  # keypoints that went through k_desc_level


@pytest.mark.parametrize("w,h,nf", [(640, 480, 1000), (752, 480, 1000), (1241, 376, 2000), (333, 257, 300), (320, 240, 500), (1920, 1080, 3000)])
def test_sizes_of_every_baseline_config(w, h, nf):
    imgs = [synth.synth_frame(30 + i, w, h) for i in range(2)] + [synth.synth_natural(40, w, h)]
    first, nreg, n = _check(imgs, nfeatures=nf)
    assert n > 0


@pytest.mark.parametrize("preset", [0, 1])
def test_both_gaussian_presets(preset):
    _check([synth.synth_frame(3), synth.synth_natural(4)], gauss=preset)


def test_saturated_content_reaches_the_blur_clamp():
    """Flat 255 areas: with the legacy taps (sum 257) the column pass exceeds 255 << 16 and saturates like cv::saturate_cast."""
    rng = np.random.default_rng(9)
    img = np.full((480, 640), 255, np.uint8)
    for _ in range(300):
        x, y = int(rng.integers(0, 600)), int(rng.integers(0, 440))
        img[y:y + int(rng.integers(4, 40)), x:x + int(rng.integers(4, 40))] = int(rng.choice([0, 255, 254, 128]))
    _check([img], gauss=0)
    _check([255 - img], gauss=1)


def test_keypoints_along_the_image_border():
    """Corners 19..22 px from every border of every level: the reflect-101 frame around a region and the first / last blur band."""
    rng = np.random.default_rng(1)
    img = np.full((480, 640), 90, np.uint8)
    for x in range(4, 640 - 30, 37):                                # blocks of every size hugging the four borders: their corners
        s = int(rng.integers(8, 30))                                # survive on the upper levels, 19..25 level pixels from the edge
        img[0:s + 18, x:x + s] = int(rng.choice([10, 230]))
        img[480 - s - 18:480, x + 5:x + 5 + s] = int(rng.choice([20, 240]))
    for y in range(4, 480 - 30, 37):
        s = int(rng.integers(8, 30))
        img[y:y + s, 0:s + 18] = int(rng.choice([15, 235]))
        img[y + 5:y + 5 + s, 640 - s - 18:640] = int(rng.choice([25, 245]))
    img += rng.integers(0, 5, img.shape).astype(np.uint8)
    first, nreg, n = _check([img, img[::-1, ::-1].copy()])
    assert n > 50
    _check([np.ascontiguousarray(img.T[:480, :480])])


@pytest.mark.parametrize("nf,levels,sf", [(500, 8, 1.2), (2000, 8, 1.2), (1000, 12, 1.1), (1000, 5, 1.5), (1500, 6, 1.3), (4000, 8, 1.2)])
def test_other_quotas_level_counts_and_scale_factors(nf, levels, sf):
    """Quotas change the per-level keypoint capacity (the workgroup's keypoint arrays), scale factors the level sizes and so
    which levels qualify and how they are tiled."""
    _check([synth.synth_frame(11, 752, 480), synth.synth_natural(12, 752, 480)], nfeatures=nf, levels=levels, sf=sf, min_regions=0)


def test_batch_equals_the_per_keypoint_kernel(monkeypatch):
    """The same 40-frame device batch through both forms of the descriptor stage (ORB_DESC_LEVEL=0: every level one wave per
    keypoint): byte-identical keypoints and descriptors, frame by frame; a sample against the oracle."""
    imgs = synth.synth_sequence(4100, 40, 640, 480)
    ex = capi.Extractor()
    got = ex.extract_batch(imgs)
    assert ex.desc_plan()[1] > 0
    monkeypatch.setenv("ORB_DESC_LEVEL", "0")
    ex0 = capi.Extractor()
    got0 = ex0.extract_batch(imgs)
    assert ex0.desc_plan() == (8, 0)
    for i in range(len(imgs)):
        assert got[i][0].tobytes() == got0[i][0].tobytes() and np.array_equal(got[i][1], got0[i][1]), i
    ref = oracle.Extractor()
    for i in (0, 13, 39):
        rk, rd = ref.extract(imgs[i])
        assert got[i][0].tobytes() == rk.tobytes() and np.array_equal(got[i][1], rd)
    ex.close(); ex0.close()


def test_single_frame_graph_replay_with_the_level_kernel():
    """orb_extract captures its chain as a graph after two calls: with the level-resident kernel in the chain the replays
    return what the oracle returns, across images."""
    imgs = [synth.synth_frame(60 + i) for i in range(3)]
    ref = oracle.Extractor()
    want = [ref.extract(im) for im in imgs]
    ex = capi.Extractor()
    for rep in range(4):
        for im, (rk, rd) in zip(imgs, want):
            k, d = ex.extract(im)
            assert k.tobytes() == rk.tobytes() and np.array_equal(d, rd), rep
    ex.close()


def test_frame_without_keypoints_on_some_levels():
    """A frame whose upper levels hold no corner at all (regions with nothing to do leave at their first barrier) next to a
    busy one in the same batch."""
    flat = np.full((480, 640), 128, np.uint8)
    flat[200:203, 300:303] = 255                                     # one tiny dot: gone after a few levels
    _check([flat, synth.synth_frame(5), np.full((480, 640), 7, np.uint8)], min_regions=1)
