#!/usr/bin/env python3
"""Run ON THE GPU BOX: where the time of the quadtree kernel (k_quadtree, one workgroup per (frame, level)) goes, per level, on
drawn shapes and on natural-statistics content: mean microseconds between the stage stamps of orb_extractor_set_qt_stamps.
  usage: tools/qt_stamps.py [frames]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))
from orbhip import capi, synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    W, H = 640, 480
    dev = torch.device("cuda", 0)
    for content in ("shapes", "natural"):
        ex = capi.Extractor(1000)
        cap = ex.max_keypoints
        imgs = synth.synth_sequence(0, min(B, 64), W, H, content=content)
        imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
        d_img = torch.from_numpy(imgs).to(dev)
        d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
        d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        run = lambda: ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, W * H, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_cnt.data_ptr())
        for _ in range(4):
            run(); ex.sync()
        d_st = torch.zeros(B * 8 * 8, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ex.set_qt_stamps(d_st.data_ptr(), d_st.numel())
        run(); ex.sync()
        ex.set_qt_stamps(0, 0)
        st = d_st.cpu().numpy().astype(np.int64).reshape(8, B, 8)
        t0, t1 = st[:, :, 0].min(), st[:, :, 4].max()
        print("%s: kernel span %.1f us; per level: candidates, list size / expandable nodes at the first careful iteration, careful iterations;"
              " us: sort, full passes, careful phase, emission | workgroup" % (content, (t1 - t0) / 100.0))
        for l in range(8):
            s = st[l]
            w5 = s[:, 5].astype(np.uint64)
            n = (w5 & np.uint64(0xFFFF)).astype(int); it = ((w5 >> np.uint64(16)) & np.uint64(0xFFFF)).astype(int)
            pc = ((w5 >> np.uint64(32)) & np.uint64(0xFFFF)).astype(int); sz = ((w5 >> np.uint64(48)) & np.uint64(0xFFFF)).astype(int)
            d = [(s[:, k + 1] - s[:, k]).mean() / 100.0 for k in range(4)]
            life = (s[:, 4] - s[:, 0]) / 100.0
            print("  level %d: %5.0f cand (max %4d), list %4.0f / %4.0f expandable, %.1f iterations | %5.2f %5.2f %5.2f %5.2f | %5.2f (p99 %.1f, max %.1f); ends %.1f us after the kernel's start"
                  % (l, n.mean(), n.max(), sz.mean(), pc.mean(), it.mean(), d[0], d[1], d[2], d[3], life.mean(), np.percentile(life, 99), life.max(),
                     (s[:, 4].max() - t0) / 100.0))
        ex.close()


if __name__ == "__main__":
    main()
