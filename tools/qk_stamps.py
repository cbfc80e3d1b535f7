#!/usr/bin/env python3
"""Run ON THE GPU BOX: the query-form SearchByBoW (orb_match_bow_query_device) of BASELINE configs[4] ALONE -- one 752x480
stream frame against the 1000-keyframe DB, nothing else on the GPU -- timed with events on the matcher's stream, plus the
per-workgroup stage stamps of orb_matcher_set_stage_stamps (where inside a workgroup the time goes).
  usage: tools/qk_stamps.py [n_kf] [repeats]"""
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))
sys.path.insert(0, ROOT)
from orbhip import capi, synth  # noqa: E402
import bench  # noqa: E402


def main():
    n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    W, H = 752, 480
    dev = torch.device("cuda", 0)
    ex, mt = capi.Extractor(1000), capi.Matcher(0.7, True)
    cap = ex.max_keypoints
    n_q = 16
    F = n_kf + n_q
    buf = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)
    d_kps, d_desc = buf(F * cap * 28, torch.uint8), buf(F * cap * 32, torch.uint8)
    d_counts, d_node = buf(F, torch.int32), buf(F * cap, torch.int16)
    d_valid = torch.from_numpy(np.stack([synth.synth_valid_flags(cap, 7000 + i) for i in range(F)])).to(dev)
    tree = bench.trained_vocabulary(ex, W, H)
    voc = capi.Vocabulary(tree)
    n_nodes = voc.level_nodes(4)
    frames = [synth.synth_sequence(k0, min(40, n_kf - k0), W, H) for k0 in range(0, n_kf, 40)]
    frames.append(np.concatenate([synth.synth_sequence(8 * ((3 + 61 * g) % (n_kf // 8)) + 3, 1, W, H, noise=5) for g in range(n_q)]))
    k0 = 0
    for fr in frames:
        n = len(fr)
        d_b = torch.from_numpy(fr).to(dev)
        ex.extract_batch_device(d_b.data_ptr(), n, H, W, W, W * H, d_kps.data_ptr() + k0 * cap * 28, d_desc.data_ptr() + k0 * cap * 32, cap,
                                d_counts.data_ptr() + k0 * 4)
        ex.sync()
        k0 += n
    d_ck, d_cs, d_cc, d_cd = buf(F * cap, torch.int32), buf(F * n_nodes, torch.int16), buf(F * n_nodes, torch.int16), buf(F * cap * 32, torch.uint8)
    voc.transform_device(mt, d_desc.data_ptr(), d_counts.data_ptr(), F, cap, 4, d_node_of=d_node.data_ptr())
    mt.build_csr_desc_device(d_node.data_ptr(), d_counts.data_ptr(), d_desc.data_ptr(), F, cap, n_nodes, d_ck.data_ptr(), d_cs.data_ptr(),
                             d_cc.data_ptr(), d_cd.data_ptr())
    mt.sync()
    store = dict(desc=d_desc.data_ptr(), kps=d_kps.data_ptr(), valid=d_valid.data_ptr(), counts=d_counts.data_ptr(),
                 node_of=d_node.data_ptr(), cap=cap, n_frames=F, n_nodes=n_nodes, csr_keys=d_ck.data_ptr(), csr_start=d_cs.data_ptr(),
                 csr_cnt=d_cc.data_ptr(), csr_desc=d_cd.data_ptr())
    kf_idx = torch.arange(n_kf, dtype=torch.int32, device=dev)
    f_idx = torch.arange(n_kf, F, dtype=torch.int32, device=dev)
    d_m, d_n = buf(n_kf * cap, torch.int32), buf(n_kf, torch.int32)
    st = torch.cuda.ExternalStream(mt.stream, device=dev)
    times = []
    for r in range(reps + 5):
        qi = r % n_q
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        mt.match_bow_query_device(store, kf_idx.data_ptr(), n_kf, f_idx.data_ptr() + 4 * qi, 1, d_m.data_ptr(), d_n.data_ptr())
        e1.record(st)
        mt.sync()
        if r >= 5:
            times.append(e0.elapsed_time(e1) * 1e3)
    print("query form alone, %d keyframes: median %.1f us  min %.1f  max %.1f  (events on the matcher stream; mean matches/pair %.1f)"
          % (n_kf, statistics.median(times), min(times), max(times), float(d_n.float().mean())))
    # stage stamps of one call
    nblk = 4096
    d_st = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    capi.lib().orb_matcher_set_stage_stamps(mt.h, d_st.data_ptr(), nblk * 8)
    mt.match_bow_query_device(store, kf_idx.data_ptr(), n_kf, f_idx.data_ptr(), 1, d_m.data_ptr(), d_n.data_ptr())
    mt.sync()
    capi.lib().orb_matcher_set_stage_stamps(mt.h, None, 0)
    s = d_st.cpu().numpy().reshape(nblk, 8)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    rel = (s[:, :7] - t0) / 100.0                                   # us (100 MHz)
    names = ["start", "query staged + sync", "phase 1 done", "phase 2 start (sync)", "phase 2 done", "finish start (sync)", "row written"]
    print("%d workgroups; stage boundaries relative to the first workgroup's start, us: median / p90 / max" % len(s))
    for k, nme in enumerate(names):
        c = rel[:, k]
        print("  %-24s %7.1f %7.1f %7.1f" % (nme, np.median(c), np.percentile(c, 90), c.max()))
    d = np.diff(rel, axis=1)
    print("stage durations per workgroup, us: median / p90 / max")
    for k in range(6):
        print("  %-24s %7.1f %7.1f %7.1f" % (names[k + 1], np.median(d[:, k]), np.percentile(d[:, k], 90), d[:, k].max()))
    nm = d_n.cpu().numpy()
    slow = np.argsort(-d[:, 3])[:8]
    print("slowest phase 2 workgroups:", [(int(b), round(float(d[b, 3]), 1), int(nm[b]) if b < n_kf else -1) for b in slow])


if __name__ == "__main__":
    main()
