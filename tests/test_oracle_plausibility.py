"""Independent plausibility checks of the oracle's OpenCV primitives.  The reference calls cv::resize,
cv::GaussianBlur, cv::fastAtan2 and cv::FAST (src/ORBextractor.cc:1166, :1130, :104, :853-861) and ships no test vectors;
oracle/ restates them from SURVEY Appendix A.  These are TOLERANCE checks against implementations written by others (torch's
bilinear interpolation, scipy's correlation, numpy's arctan2, scikit-image's FAST) -- they catch the gross class of restatement
error (pixel-centre convention, tap order, border mode, angle octant, ring geometry), they do NOT pin the oracle to OpenCV:
parity stays "unpinned" (DESIGN.md section 6)."""
import os

import numpy as np
import pytest

import oracle
from orbhip import synth

GOLD2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_crosscheck2.npz"))


def _images():
    return [synth.synth_frame(0, 320, 240), synth.synth_natural(3, 320, 240),
            np.random.default_rng(5).integers(0, 256, (201, 333)).astype(np.uint8)]


@pytest.mark.parametrize("dw,dh", [(267, 200), (533, 400), (300, 150), (161, 121)])
def test_resize_is_bilinear_with_half_pixel_centres(dw, dh):
    """cv::resize INTER_LINEAR samples at (dx + 0.5) * scale - 0.5 with clamped edges -- what
    torch.nn.functional.interpolate(mode='bilinear', align_corners=False) computes in float.  The oracle's 11-bit fixed-point
    coefficients and its two truncating shifts stay within one grey level of it."""
    import torch
    for img in _images():
        want = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float32))[None, None], size=(dh, dw), mode="bilinear",
                                               align_corners=False)[0, 0].numpy().astype(np.float64)
        got = oracle.resize(img, dw, dh).astype(np.float64)
        d = np.abs(got - want)
        assert d.max() <= 1.0 + 1e-3, (img.shape, dw, dh, d.max())
        assert d.mean() < 0.4, d.mean()
        # the rounding is unbiased to a fraction of a grey level (a shifted sampling grid would show here on the ramps below)
        assert abs((got - want).mean()) < 0.25


def test_resize_sampling_grid_on_ramps():
    """A horizontal and a vertical ramp locate the sampling grid to a fraction of a pixel: a centre convention that is off by
    half a source pixel shifts the result by ~half a ramp step everywhere."""
    import torch
    w, h, dw, dh = 240, 200, 200, 167
    for img in (np.tile((np.arange(w) % 256).astype(np.uint8), (h, 1)), np.tile((np.arange(h) % 256).astype(np.uint8)[:, None], (1, w))):
        want = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float32))[None, None], size=(dh, dw), mode="bilinear",
                                               align_corners=False)[0, 0].numpy()
        got = oracle.resize(img, dw, dh).astype(np.float32)
        assert np.abs(got - want).max() <= 1.0 + 1e-3
        assert abs(float((got - want).mean())) < 0.2


@pytest.mark.parametrize("preset", [0, 1])
def test_blur_is_a_7x7_sigma2_gaussian_reflect101(preset):
    """Both integer kernels ({18,34,49,55}: sum 257, {18,34,48,56}: sum 256) against the float 7-tap sigma-2 Gaussian of
    cv::getGaussianKernel with scipy's 'mirror' border (= BORDER_REFLECT_101), on shapes, natural content and noise."""
    from scipy import ndimage
    taps = [[18, 34, 49, 55], [18, 34, 48, 56]][preset]
    gain = (2 * (taps[0] + taps[1] + taps[2]) + taps[3]) / 256.0
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-x * x / 8.0)
    k /= k.sum()
    for img in _images():
        want = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
        want = np.minimum(want * gain * gain, 255.0)
        got = oracle.blur(img, taps).astype(np.float64)
        d = np.abs(got - want)
        # an 8.8 tap deviates from the float kernel by up to 0.5 / 256 (0.82 / 256 for the error-diffused 48): a grey level or
        # two on full-contrast noise, a fraction of one on average
        assert d.max() <= 2.0 and d.mean() < 0.45, (preset, d.max(), d.mean())
    # an impulse shows tap ORDER and symmetry exactly: the response is the outer product of the taps
    imp = np.zeros((31, 31), np.uint8)
    imp[15, 15] = 255
    got = oracle.blur(imp, taps).astype(np.int64)
    t = np.array(taps + taps[2::-1], np.int64)
    want = (255 * np.outer(t, t) + 32768) >> 16
    assert np.array_equal(got[12:19, 12:19], want) and got.sum() == want.sum()


def test_ic_angle_is_the_atan2_of_the_patch_moments():
    """IC_Angle (:78-105) + cv::fastAtan2 against numpy: moments over the 31x31 disc (|u| <= umax[|v|]) in float64, arctan2 in
    degrees; fastAtan2's degree-7 polynomial is documented to ~0.3 degrees."""
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    vv, uu = np.mgrid[-15:16, -15:16]
    disc = np.abs(uu) <= np.array(umax)[np.abs(vv)]
    assert disc.sum() == 749
    worst = 0.0
    for img in _images():
        ref = oracle.Extractor(500, 1.2, 1, 20, 7)
        ref.compute_pyramid(img)
        h, w = img.shape
        rng = np.random.default_rng(11)
        for _ in range(150):
            x, y = int(rng.integers(16, w - 16)), int(rng.integers(16, h - 16))
            p = img[y - 15:y + 16, x - 15:x + 16].astype(np.float64)
            m10, m01 = float((uu * p)[disc].sum()), float((vv * p)[disc].sum())
            if abs(m10) + abs(m01) < 50:
                continue                                            # flat patch: the angle is ill-defined
            want = np.degrees(np.arctan2(m01, m10)) % 360.0
            got = ref.ic_angle(0, x, y)
            worst = max(worst, abs((got - want + 180.0) % 360.0 - 180.0))
    assert worst < 0.3, worst


def test_fast_atan2_octants_and_axes():
    """The octant logic of cv::fastAtan2 on the axes and diagonals (exact values of the polynomial are KATs elsewhere)."""
    for y, x, want in [(0, 1, 0), (1, 1, 45), (1, 0, 90), (1, -1, 135), (0, -1, 180), (-1, -1, 225), (-1, 0, 270), (-1, 1, 315)]:
        got = oracle.fast_atan2(float(y) * 1000, float(x) * 1000)
        assert abs((got - want + 180.0) % 360.0 - 180.0) < 0.3, (y, x, got)
    assert oracle.fast_atan2(0.0, 0.0) == 0.0


@pytest.mark.parametrize("name", ["nat_a", "nat_b"])
def test_fast9_masks_on_natural_content_equal_skimage(name):
    """The FAST-9/16 detection mask (V > t) against scikit-image's corner_fast on frames with natural image
    statistics, at seven thresholds from 5 to 60: identical masks."""
    img = GOLD2[name]
    h, w = img.shape
    v = oracle.fast_vmap(img)
    inner = np.zeros((h, w), bool)
    inner[3:h - 3, 3:w - 3] = True
    total = 0
    for t in (5, 7, 12, 20, 30, 45, 60):
        mask = np.unpackbits(GOLD2["%s_fast9_t%d" % (name, t)])[:h * w].reshape(h, w).astype(bool)
        ours = (v > t) & inner
        assert np.array_equal(ours, mask), "threshold %d: %d vs %d corners" % (t, ours.sum(), mask.sum())
        total += int(mask.sum())
    assert total > 500


@pytest.mark.parametrize("name", ["nat_a", "nat_b"])
def test_orientation_on_natural_content_matches_skimage(name):
    img = GOLD2[name]
    ref = oracle.Extractor(500, 1.2, 1, 20, 7)
    ref.compute_pyramid(img)
    pts = GOLD2[name + "_points_rc"]
    want = np.degrees(GOLD2[name + "_angles_rad"]) % 360.0
    got = np.array([ref.ic_angle(0, int(c), int(r)) for r, c in pts], np.float64)
    d = np.abs((got - want + 180.0) % 360.0 - 180.0)
    assert d.max() < 0.35, d.max()
