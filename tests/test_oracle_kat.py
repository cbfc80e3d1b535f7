"""Known-answer tests pinning the CPU oracle to facts derivable from the reference source alone
(SURVEY.md Appendix C).  The reference ships no fixtures, so these are the only external pins."""
import hashlib
import os
import re
import struct

import numpy as np

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_brief_pattern_sha256():
    # sha256 of the int32-LE serialisation of bit_pattern_31_ (src/ORBextractor.cc:175-432)
    txt = open(os.path.join(ROOT, "include", "orb_brief_pattern.h")).read()
    body = txt.split("{", 1)[1].split("}", 1)[0]
    vals = [int(v) for v in re.findall(r"-?\d+", body)]
    assert len(vals) == 1024
    assert min(vals) == -13 and max(vals) == 12
    assert hashlib.sha256(struct.pack("<1024i", *vals)).hexdigest() == \
        "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"


def test_ctor_tables_1000():
    t = oracle.Extractor(1000, 1.2, 8, 20, 7).tables()
    assert t["quota"].tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert 1 + 2 * sum(2 * u + 1 for u in t["umax"][1:]) + 2 * 15 == 749      # patch pixels
    want = [1, 1.2, 1.44, 1.728, 2.0736, 2.48832, 2.985985, 3.583182]
    assert np.allclose(t["scale"], want, rtol=2e-7 * 8)
    assert [int(np.float32(31) * s) for s in t["scale"]] == [31, 37, 44, 53, 64, 77, 92, 111]
    assert np.array_equal(t["inv_scale"], np.float32(1) / t["scale"])
    assert np.array_equal(t["sigma2"], t["scale"] * t["scale"])


def test_ctor_tables_2000():
    t = oracle.Extractor(2000, 1.2, 8, 20, 7).tables()
    assert t["quota"].tolist() == [434, 362, 302, 251, 209, 175, 145, 122]


def _dims(cols, rows):
    e = oracle.Extractor()
    e.compute_pyramid(np.zeros((rows, cols), np.uint8))
    return [e.pyramid_level(l).shape[::-1] for l in range(8)]


def test_pyramid_dims():
    d = _dims(640, 480)
    assert d == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    assert sum(w * h for w, h in d) == 950532
    d = _dims(752, 480)
    assert d == [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]
    assert sum(w * h for w, h in d) == 1117367
    d = _dims(1241, 376)
    assert d == [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]
    assert sum(w * h for w, h in d) == 1444097


def test_resize_coefficients_640_to_533():
    ofs, c0, c1 = oracle.resize_tab(640, 533)
    assert list(zip(ofs[:4].tolist(), c0[:4].tolist(), c1[:4].tolist())) == \
        [(0, 1842, 206), (1, 1431, 617), (2, 1020, 1028), (3, 609, 1439)]
    assert (int(ofs[-1]), int(c0[-1]), int(c1[-1])) == (638, 206, 1842)
    assert np.all(c0.astype(int) + c1.astype(int) == 2048)


def test_resize_constant_and_ramp():
    src = np.full((48, 64), 77, np.uint8)
    assert np.all(oracle.resize(src, 53, 40) == 77)
    ramp = np.tile(np.arange(64, dtype=np.uint8) * 4, (48, 1))
    out = oracle.resize(ramp, 53, 40)
    assert np.all(np.diff(out.astype(int), axis=1) >= 0) and np.all(out[0] == out[-1])


def test_gaussian_taps_via_impulse():
    src = np.zeros((21, 21), np.uint8)
    src[10, 10] = 255
    out = oracle.blur(src).astype(int)
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    want = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(out[7:14, 7:14], want)
    assert out.sum() == want.sum()
    assert np.all(oracle.blur(np.full((9, 30), 200, np.uint8)) == ((200 * 257 * 257 + 32768) >> 16))


def test_gaussian_reflect101_border():
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (12, 15), dtype=np.uint8)
    pad = np.pad(src, 3, mode="reflect")           # numpy 'reflect' == BORDER_REFLECT_101
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    rows = sum(k[t] * pad[:, t:t + 15].astype(np.int64) for t in range(7))
    full = sum(k[t] * rows[t:t + 12] for t in range(7))
    want = np.clip((full + 32768) >> 16, 0, 255)
    assert np.array_equal(oracle.blur(src), want.astype(np.uint8))


def test_fast_atan2_constants_and_axes():
    f = np.float32
    p1 = f(0.9997878412794807) * f(180 / np.pi)
    assert p1.view(np.uint32) == 0x4265226F
    assert (f(-0.3258083974640975) * f(180 / np.pi)).view(np.uint32) == 0xC19556EE
    assert (f(0.1555786518463281) * f(180 / np.pi)).view(np.uint32) == 0x410E9FBF
    assert (f(-0.04432655554792128) * f(180 / np.pi)).view(np.uint32) == 0xC0228AD9
    assert oracle.fast_atan2(0, 0) == 0
    assert oracle.fast_atan2(0, 5) == 0
    assert abs(oracle.fast_atan2(5, 0) - 90) < 1e-4
    assert abs(oracle.fast_atan2(0, -5) - 180) < 1e-4
    assert abs(oracle.fast_atan2(-5, 0) - 270) < 1e-4
    for y, x in [(3, 4), (-7, 2), (100, -1), (-1, -100), (12345, 54321)]:
        assert abs(float(oracle.fast_atan2(y, x)) - (np.degrees(np.arctan2(y, x)) % 360)) < 0.02


def test_cvround_half_even():
    assert [oracle.cvround(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_fast_score_hand_patterns():
    # bright centre on dark ring: every 9-arc has min diff 100 -> V = 100
    im = np.full((7, 7), 50, np.uint8)
    im[3, 3] = 150
    assert oracle.fast_vmap(im)[3, 3] == 100
    # dark centre: V = 100 via the "darker" branch
    im = np.full((7, 7), 150, np.uint8)
    im[3, 3] = 50
    assert oracle.fast_vmap(im)[3, 3] == 100
    # exactly 9 contiguous ring pixels darker by 30, the other 7 equal to the centre: V = 30
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3),
            (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    for start in range(16):
        im = np.full((7, 7), 100, np.uint8)
        for t in range(9):
            dx, dy = ring[(start + t) % 16]
            im[3 + dy, 3 + dx] = 70
        assert oracle.fast_vmap(im)[3, 3] == 30
        # only 8 contiguous: not a corner for any threshold >= 0
        dx, dy = ring[(start + 8) % 16]
        im[3 + dy, 3 + dx] = 100
        assert oracle.fast_vmap(im)[3, 3] <= 0


def test_fast_cells_counts_640x480():
    # Appendix C: 815 FAST cells at 640x480 -- checked indirectly: a frame of isolated bright dots,
    # one per 40 px, yields exactly one candidate per dot inside the detection band.
    im = np.full((480, 640), 20, np.uint8)
    ys, xs = np.mgrid[40:440:40, 40:600:40]
    im[ys, xs] = 220
    e = oracle.Extractor()
    e.compute_pyramid(im)
    c = e.cell_candidates(0)
    got = {(int(x) + 16, int(y) + 16) for x, y, _ in c}
    assert got == {(int(x), int(y)) for x, y in zip(xs.ravel(), ys.ravel())}
    assert set(c[:, 2].tolist()) == {199}          # V = 200 -> score 199
