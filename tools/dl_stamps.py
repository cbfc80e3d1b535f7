#!/usr/bin/env python3
"""Run ON THE GPU BOX: where the time of the level-resident descriptor kernel (k_desc_level, csrc/orb_desc_level.hip) goes.
A device batch of N 640x480 frames (default 512) is extracted a few times with orb_extractor_set_desc_stamps on; per region of
the handle's plan the table lists the mean microseconds between the phase stamps of its workgroups (100 MHz clock) and the
span of the whole launch (first start .. last end over all workgroups).
  usage: tools/dl_stamps.py [frames] [width height nfeatures]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))
from orbhip import capi, synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    W, H, NF = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (640, 480, 1000)
    dev = torch.device("cuda", 0)
    ex = capi.Extractor(NF)
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
    first, nreg = ex.desc_plan()
    print("plan: first level %d, %d regions; %d frames" % (first, nreg, B))
    if nreg == 0:
        return
    d_st = torch.zeros(B * nreg * 8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ex.set_desc_stamps(d_st.data_ptr(), d_st.numel())
    run(); ex.sync()
    st = d_st.cpu().numpy().astype(np.uint64).reshape(B * nreg, 8)
    ex.set_desc_stamps(0, 0)
    live = st[:, 6] > 0
    t0 = st[st[:, 0] > 0, 0].min()
    print("launch span: %.1f us over %d workgroups (%d without keypoints)" % ((st[live, 6].max() - t0) / 100.0, len(st), int((~live).sum())))
    names = ["stage", "fix+IC", "(barrier)", "angles", "blur", "(barrier)+write", "sample"]
    print("%-8s %5s %5s | %s | total" % ("region", "wgs", "kps", " ".join("%9s" % n for n in names)))
    reg = (st[:, 7] >> np.uint64(32)).astype(int)
    nk = (st[:, 7] & np.uint64(0xFFFFFFFF)).astype(int)
    ends = []
    for r in range(nreg):
        m = live & (reg == r)
        if not m.any():
            continue
        s = st[m].astype(np.int64)
        # stamps: 0 start, 1 staged, 2 IC done, 3 after the barrier + angles, 4 blur computed, 5 written (after two barriers), 6 end
        d = [(s[:, 1] - s[:, 0]), (s[:, 2] - s[:, 1]), None, (s[:, 3] - s[:, 2]), (s[:, 4] - s[:, 3]), (s[:, 5] - s[:, 4]), (s[:, 6] - s[:, 5])]
        cells = " ".join("%9s" % ("" if x is None else "%.2f" % (x.mean() / 100.0)) for x in d)
        print("%-8d %5d %5.0f | %s | %.2f" % (r, int(m.sum()), nk[m].mean(), cells, (s[:, 6] - s[:, 0]).mean() / 100.0))
    # how many workgroups are in flight over the launch (x 8 waves): the occupancy the LDS and registers allow
    s = st[live].astype(np.int64)
    dur = (s[:, 6] - s[:, 0]).sum() / 100.0
    print("sum of workgroup lifetimes %.0f us = %.1f workgroups in flight on average over the span" % (dur, dur / ((st[live, 6].max() - t0) / 100.0)))


if __name__ == "__main__":
    main()
