#!/usr/bin/env python3
"""Single-frame latency of the drop-in path (host image in, host keypoints/descriptors out), the way Frame::ExtractORB
calls it: one orb_extract per frame.  Prints median / p90 milliseconds over N frames."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/orb-slam2-chinesenotes_amd/pyhost")
import torch  # noqa: F401  (one HIP runtime per process: torch first)
from orbhip import capi, synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    import os
    print("HIP graph replay: %s" % ("off (ORB_NO_GRAPH)" if os.environ.get("ORB_NO_GRAPH") else "on"))
    for (w, h, nf) in ((640, 480, 1000), (752, 480, 1000), (1241, 376, 2000)):
        ex = capi.Extractor(nf, 1.2, 8, 20, 7)
        imgs = [synth.synth_frame(i, w, h) for i in range(8)]
        for i in range(10):
            ex.extract(imgs[i % 8])
        ts = []
        for i in range(n):
            t0 = time.perf_counter()
            ex.extract(imgs[i % 8])
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = np.array(ts)
        print("%dx%d nfeatures=%d: median %.3f ms  p90 %.3f ms  min %.3f ms" % (w, h, nf, np.median(ts), np.percentile(ts, 90), ts.min()))
        ex.close()


if __name__ == "__main__":
    main()
