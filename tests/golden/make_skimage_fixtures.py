#!/opt/conda/bin/python3.9
"""Independent cross-check vectors from scikit-image 0.18.3 (NOT the reference, and not OpenCV): run in the build
container with /opt/conda/bin/python3.9, writes tests/golden/skimage_crosscheck.npz.  They pin, against a third
implementation, three definitions the oracle shares with OpenCV: the FAST-9/16 segment test (ring offsets, strict
inequalities, 9 contiguous pixels), the intensity-centroid moments over the 31x31 circular patch (umax table), and
the rBRIEF sampling pattern.  Fixtures hold data only (an input image, masks, angles, the pattern)."""
import os

import numpy as np
from skimage.feature import corner_fast, corner_orientations
from skimage.feature.orb import OFAST_MASK
import skimage


def synth(rng, h, w):
    img = np.full((h, w), 128, np.int32)
    for _ in range(60):
        x0, y0 = rng.integers(0, w), rng.integers(0, h)
        sw, sh = rng.integers(4, 60), rng.integers(4, 60)
        img[y0:y0 + sh, x0:x0 + sw] = rng.integers(0, 256)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(30):
        cx, cy, r = rng.integers(0, w), rng.integers(0, h), rng.integers(3, 25)
        img[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = rng.integers(0, 256)
    img += rng.integers(-6, 7, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    rng = np.random.default_rng(20261003)
    img = synth(rng, 120, 160)
    f = img.astype(np.float64)                       # integer-valued floats: the threshold compares are exact
    out = {"image": img}
    for t in (7, 20, 40):
        resp = corner_fast(f, n=9, threshold=float(t))
        out["fast9_mask_t%d" % t] = np.packbits(resp > 0)
    # orientation at a grid of interior points (31x31 mask needs 15 px of margin)
    pts = np.array([(r, c) for r in range(16, 104, 11) for c in range(16, 144, 13)], np.intp)
    out["orient_points_rc"] = pts.astype(np.int32)
    out["orient_angles_rad"] = corner_orientations(f, pts, OFAST_MASK).astype(np.float64)
    pos = np.loadtxt(os.path.join(os.path.dirname(skimage.__file__), "feature", "orb_descriptor_positions.txt"), dtype=np.int8)
    out["orb_positions"] = pos                       # 256 x 4
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "skimage_crosscheck.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
