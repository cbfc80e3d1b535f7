"""Independent cross-check of the oracle against scikit-image 0.18.3 (a third implementation; vectors generated in
the build container by tests/golden/make_skimage_fixtures.py).  This does NOT pin the oracle to the reference's
OpenCV -- parity stays "unpinned"  -- but it checks three shared definitions against code that was
written by someone else: the FAST-9/16 segment test, the intensity-centroid moments over the circular 31x31 patch,
and the rBRIEF sampling pattern."""
import os

import numpy as np

import oracle

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_crosscheck.npz"))


def test_fast9_corner_masks_equal_skimage():
    img = GOLD["image"]
    h, w = img.shape
    v = oracle.fast_vmap(img)                                 # V(p): corner <=> V > t
    inner = np.zeros((h, w), bool)
    inner[3:h - 3, 3:w - 3] = True
    for t in (7, 20, 40):
        mask = np.unpackbits(GOLD["fast9_mask_t%d" % t])[:h * w].reshape(h, w).astype(bool)
        ours = (v > t) & inner
        assert mask[~inner].sum() == 0
        assert np.array_equal(ours, mask), "FAST-9 mask differs at threshold %d (%d vs %d corners)" % (t, ours.sum(), mask.sum())
        assert mask.sum() > 50                                # the vector is not trivial


def test_intensity_centroid_angle_matches_skimage():
    img = GOLD["image"]
    ref = oracle.Extractor(500, 1.2, 1, 20, 7)
    ref.compute_pyramid(img)
    pts = GOLD["orient_points_rc"]
    want = np.degrees(GOLD["orient_angles_rad"]) % 360.0
    got = np.array([ref.ic_angle(0, int(c), int(r)) for r, c in pts], np.float64)
    d = np.abs((got - want + 180.0) % 360.0 - 180.0)
    # cv::fastAtan2 is a degree-7 polynomial: documented accuracy ~0.3 degrees
    assert d.max() < 0.35, "max orientation difference %.4f deg" % d.max()


def test_brief_pattern_equals_skimage_copy():
    from orbhip import capi
    ours = capi.builtin_pattern().reshape(256, 4)
    assert np.array_equal(ours, GOLD["orb_positions"])


def test_blur_is_a_sigma2_gaussian_with_mirror_border_scipy():
    # float 7-tap Gaussian (sigma 2, normalised, what cv::getGaussianKernel(7, 2) returns) with scipy's 'mirror'
    # (= BORDER_REFLECT_101): the 8.8 fixed-point oracle must stay within rounding of it
    from scipy import ndimage
    img = GOLD["image"]
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-x * x / 8.0)
    k /= k.sum()
    want = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    # the 8.8 taps 18,34,49,55 sum to 257, not 256 (each tap is rounded on its own): gain (257/256)^2
    want *= (257.0 / 256.0) ** 2
    got = oracle.blur(img).astype(np.float64)
    d = np.abs(got - np.minimum(want, 255.0))
    assert d.max() <= 1.5 and d.mean() < 0.5, (d.max(), d.mean())


def test_resize_is_pixel_centre_bilinear_scipy():
    # independent float bilinear sampling at src = (dst + 0.5) * scale - 0.5 with edge clamping
    from scipy import ndimage
    img = GOLD["image"]
    h, w = img.shape
    dw, dh = int(round(w / 1.2)), int(round(h / 1.2))
    ys = (np.arange(dh) + 0.5) * (h / dh) - 0.5
    xs = (np.arange(dw) + 0.5) * (w / dw) - 0.5
    yy, xx = np.meshgrid(ys, xs, indexing="ij")
    want = ndimage.map_coordinates(img.astype(np.float64), [yy, xx], order=1, mode="nearest")
    got = oracle.resize(img, dw, dh).astype(np.float64)
    d = np.abs(got - want)
    assert d.max() <= 1.0 and d.mean() < 0.3, (d.max(), d.mean())
