#!/opt/conda/bin/python3.9
"""More cross-check vectors from scikit-image 0.18.3 (a third implementation -- NOT the reference, not OpenCV), on content with
natural image statistics and at more thresholds: run in the build container with /opt/conda/bin/python3.9,
writes tests/golden/skimage_crosscheck2.npz.  Data only: two seeded input images (orbhip.synth.synth_natural, the generator of
the natural-content parity tests), packed FAST-9/16 corner masks at seven thresholds each, intensity-centroid orientations at a
grid of points."""
import os
import sys

import numpy as np
from skimage.feature import corner_fast, corner_orientations
from skimage.feature.orb import OFAST_MASK

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "orb-slam2-chinesenotes_amd", "pyhost", "orbhip"))
import synth  # noqa: E402  (the module alone: the package's __init__ would load the HIP library)

THRESHOLDS = (5, 7, 12, 20, 30, 45, 60)


def main():
    out = {}
    for name, seed, w, h in (("nat_a", 501, 208, 160), ("nat_b", 502, 176, 144)):
        img = synth.synth_natural(seed, w, h)
        f = img.astype(np.float64)
        out[name] = img
        for t in THRESHOLDS:
            out["%s_fast9_t%d" % (name, t)] = np.packbits(corner_fast(f, n=9, threshold=float(t)) > 0)
        pts = np.array([(r, c) for r in range(16, h - 16, 9) for c in range(16, w - 16, 11)], np.intp)
        out[name + "_points_rc"] = pts.astype(np.int32)
        out[name + "_angles_rad"] = corner_orientations(f, pts, OFAST_MASK).astype(np.float64)
    dst = os.path.join(HERE, "skimage_crosscheck2.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
