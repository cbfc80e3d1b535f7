"""The C++ oracle against tests/spec_numpy.py -- a second, structurally different implementation of SURVEY Appendix A
(written from the appendix's text) -- on fuzzed inputs, bit for bit: resize (A.2), blur (A.7), FAST score map, the cell loop
with cell-local NMS and the threshold fallback (A.4), the quadtree (A.6).  Agreement does not pin the oracle to OpenCV (both
follow the same written specification; "parity unpinned" stands); disagreement is a transcription bug in one of the two."""
import numpy as np
import pytest

import oracle
import spec_numpy as spec
from orbhip import synth


def _images(rng, n, lo=40, hi=260):
    out = []
    for i in range(n):
        w, h = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
        kind = i % 4
        if kind == 0:
            im = rng.integers(0, 256, (h, w), dtype=np.uint8)                      # white noise: every rounding path
        elif kind == 1:
            im = synth.synth_frame(int(rng.integers(0, 1000)), w, h)
        elif kind == 2:
            im = synth.synth_natural(int(rng.integers(0, 1000)), w, h)
        else:
            im = (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)               # saturated checker noise
        out.append(im)
    return out


def test_resize_equals_the_specification():
    rng = np.random.default_rng(20261004)
    for im in _images(rng, 40):
        h, w = im.shape
        for _ in range(3):
            s = float(rng.choice([1.2, 1.2, 1.1, 1.5, 2.0, 2.7, 0.8]))
            dw, dh = max(1, int(round(w / s))), max(1, int(round(h / s * float(rng.uniform(0.9, 1.1)))))
            assert np.array_equal(oracle.resize(im, dw, dh), spec.resize(im, dw, dh)), (w, h, dw, dh)
    # the sizes of the pyramid chain of BASELINE's configurations, level to level
    for (w, h) in ((640, 480), (752, 480), (1241, 376)):
        im = synth.synth_natural(5, w, h)
        sf = np.float32(1.0)
        for l in range(1, 8):
            sf = np.float32(sf * np.float64(np.float32(1.2)))
            dw, dh = int(np.rint(np.float32(w) * (np.float32(1.0) / sf))), int(np.rint(np.float32(h) * (np.float32(1.0) / sf)))
            nxt = oracle.resize(im, dw, dh)
            assert np.array_equal(nxt, spec.resize(im, dw, dh)), (w, h, l)
            im = nxt


def test_blur_equals_the_specification():
    rng = np.random.default_rng(7)
    for im in _images(rng, 40, 8, 200):
        assert np.array_equal(oracle.blur(im), spec.blur(im)), im.shape


def test_blur_with_the_error_diffused_kernel_equals_the_specification():
    """The other OpenCV integer kernel (orb_gaussian_preset 1: {18, 34, 48, 56}, sum 256) through the same arithmetic; it does
    differ from the default in low bits -- which is why the preset has to match the OpenCV the reference is built with."""
    rng = np.random.default_rng(8)
    ed = (18, 34, 48, 56)
    differ = 0
    for im in _images(rng, 30, 8, 200):
        got = oracle.blur(im, ed)
        assert np.array_equal(got, spec.blur(im, ed)), im.shape
        differ += int((got != oracle.blur(im)).sum())
    assert differ > 100


def test_fast_score_map_equals_the_specification():
    rng = np.random.default_rng(11)
    for im in _images(rng, 24, 8, 160):
        v = oracle.fast_vmap(im)
        h, w = im.shape
        assert np.array_equal(v[3:h - 3, 3:w - 3], spec.fast_v(im)), im.shape


@pytest.mark.parametrize("ini,mn", [(20, 7), (35, 12), (7, 7)])
def test_cell_loop_equals_the_specification(ini, mn):
    rng = np.random.default_rng(100 + ini)
    ims = _images(rng, 10, 70, 330) + [synth.synth_frame(3, 640, 480), synth.synth_natural(3, 400, 300)]
    for im in ims:
        ex = oracle.Extractor(500, 1.2, 3, ini, mn)
        ex.compute_pyramid(im)
        for l in range(3):
            lvl = ex.pyramid_level(l)
            want = ex.cell_candidates(l)
            got = spec.cell_candidates(lvl, ini, mn)
            assert got.shape == want.shape and np.array_equal(got, want), (im.shape, l, len(want), len(got))


def test_quadtree_equals_the_specification():
    rng = np.random.default_rng(5)
    ex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    total = 0
    for t in range(120):
        bw, bh = int(rng.integers(20, 700)), int(rng.integers(20, 500))
        if rng.random() < 0.3:
            bw = int(bh * rng.uniform(1.6, 4.4))                                  # several roots
        n = int(rng.integers(1, 1500))
        N = int(rng.integers(1, 400))
        if rng.random() < 0.5:                                                     # clustered: deep subdivision, the careful phase
            cx, cy = rng.integers(0, bw, 6), rng.integers(0, bh, 6)
            k = rng.integers(0, 6, n)
            xs = np.clip(cx[k] + rng.normal(0, bw / 25 + 1, n), 0, bw - 1).astype(np.int32)
            ys = np.clip(cy[k] + rng.normal(0, bh / 25 + 1, n), 0, bh - 1).astype(np.int32)
        else:
            xs, ys = rng.integers(0, bw, n).astype(np.int32), rng.integers(0, bh, n).astype(np.int32)
        # the cell loop emits every position at most once; responses tie often (first maximum wins)
        xy = np.unique(np.stack([ys, xs], 1), axis=0)
        rng.shuffle(xy)
        cands = np.stack([xy[:, 1], xy[:, 0], rng.integers(7, 40, len(xy))], 1).astype(np.int32)
        want = ex.distribute(cands, 16, 16 + bw, 16, 16 + bh, N)
        got = spec.distribute(cands, 16, 16 + bw, 16, 16 + bh, N)
        assert got.shape == want.shape and np.array_equal(got, want), (t, bw, bh, len(cands), N, len(want), len(got))
        total += len(want)
    assert total > 5000
