#!/usr/bin/env python3
"""Run ON THE GPU BOX: where the time of the pyramid launches (k_pyr_chain, csrc/orb_extract_kernels.hip) goes.  A device batch
of N 640x480 frames (default 512) with orb_extractor_set_pyr_stamps on; per launch the mean microseconds a workgroup spends
between its phase stamps (thread 0, 100 MHz clock), the launch's span and how many workgroups are in flight on average.
  usage: tools/pyr_stamps.py [frames] [width height]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))
from orbhip import capi, synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
    dev = torch.device("cuda", 0)
    ex = capi.Extractor(1000)
    cap = ex.max_keypoints
    imgs = synth.synth_sequence(0, min(B, 64), W, H)
    imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
    d_img = torch.from_numpy(imgs).to(dev)
    d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    run = lambda: ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, W * H, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_cnt.data_ptr())
    for _ in range(3):
        run(); ex.sync()
    d_st = torch.zeros(B * 64 * 8 * 4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ex.set_pyr_stamps(d_st.data_ptr(), d_st.numel())
    run(); ex.sync()
    ex.set_pyr_stamps(0, 0)
    st = d_st.cpu().numpy().astype(np.int64)
    off = 0
    print("%d frames of %dx%d; microseconds per workgroup (thread 0): stage = requests + stores, wait = barrier behind them, Lk = level k of the chain incl. its barrier" % (B, W, H))
    for c, (bands, steps) in enumerate(ex.pyr_stamp_layout()):
        n = bands * B
        s = st[off:off + n * 8].reshape(n, 8)
        off += n * 8
        if not (s[:, 0] > 0).all():
            print("launch %d: no stamps (capacity)" % c)
            continue
        end = s[:, 2 + steps]
        span = (end.max() - s[:, 0].min()) / 100.0
        life = (end - s[:, 0]) / 100.0
        cols = ["stage %.2f" % ((s[:, 1] - s[:, 0]).mean() / 100.0), "wait %.2f" % ((s[:, 2] - s[:, 1]).mean() / 100.0)]
        for k in range(steps):
            cols.append("L%d %.2f" % (k, (s[:, 3 + k] - s[:, 2 + k]).mean() / 100.0))
        print("launch %d: %3d bands x %d frames, %d levels | %s | workgroup %.2f us | span %.1f us | %.0f workgroups in flight (%.2f per CU)"
              % (c, bands, B, steps, "  ".join(cols), life.mean(), span, life.sum() / span, life.sum() / span / 256.0))
        # by band (first / middle / last): bands differ in rows
        for b in sorted({0, bands // 2, bands - 1}):
            sb = s[b::bands]
            print("     band %2d: workgroup %.2f us" % (b, ((sb[:, 2 + steps] - sb[:, 0]) / 100.0).mean()))


if __name__ == "__main__":
    main()
