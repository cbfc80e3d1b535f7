#!/usr/bin/env python3
"""Run ON THE GPU BOX.  How much of k_fast_strips_p is the work on pairs / candidates that only the minTh fallback needs?
Stage times of a 512-frame batch with (iniTh, minTh) = (20, 7) -- the product -- and (20, 20): the same detector with the
fallback threshold raised to iniTh, i.e. phases B and the tail see only what a cell with a keypoint at iniTh needs.  (20, 20)
gives DIFFERENT keypoints; only its FAST time is of interest: it brackets what a two-stage scheme (exact scores for the
7 < U <= 20 pairs only in cells that turn out to have no keypoint at 20) can save."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))
import torch
from orbhip import capi, synth

def run(imgs, ini, mn, reps=12):
    n, h, w = imgs.shape
    ex = capi.Extractor(1000, 1.2, 8, ini, mn)
    cap = ex.max_keypoints
    d = torch.from_numpy(imgs).cuda()
    k = torch.zeros(n * cap * 28, dtype=torch.uint8, device="cuda")
    de = torch.zeros(n * cap * 32, dtype=torch.uint8, device="cuda")
    c = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        ex.extract_batch_device(d.data_ptr(), n, h, w, w, w * h, k.data_ptr(), de.data_ptr(), cap, c.data_ptr())
    ex.sync()
    ex.set_profiling(True)
    for _ in range(reps):
        ex.extract_batch_device(d.data_ptr(), n, h, w, w, w * h, k.data_ptr(), de.data_ptr(), cap, c.data_ptr())
    ex.sync()
    ms = ex.stage_ms()
    ex.set_profiling(False)
    kp = float(c.float().mean())
    ex.close()
    return ms, kp

if __name__ == "__main__":
    n = 512
    for name, gen in (("synthetic", synth.synth_frame), ("natural", synth.synth_natural)):
        base = [gen(100 + i) for i in range(32)]
        imgs = np.stack([base[i % 32] for i in range(n)])
        for ini, mn in ((20, 7), (20, 20)):
            ms, kp = run(imgs, ini, mn)
            print("%-9s iniTh %2d minTh %2d: stages %s  keypoints/frame %.0f" % (name, ini, mn, ms, kp), flush=True)
