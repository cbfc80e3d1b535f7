#!/usr/bin/env python3
"""Timings of BASELINE.json configurations 3 and 5 on one GPU (their parity is covered by tests/test_gpu_stereo.py
and tests/test_gpu_matcher.py; this script only measures).

  config 3: KITTI-sized stereo pair 1241x376, nFeatures 2000: extract left + right (two handles, as the
            reference's two threads, src/Frame.cc:82-85) + ComputeStereoMatches on the device-resident pyramids.
            Host images in, host mvuRight / mvDepth out.
  config 5: 752x480 stream: per frame extract + vocabulary assignment + SearchByBoW against a 1000-keyframe
            descriptor DB resident in HBM (1000 (keyframe, frame) pairs per frame), device-resident.
Prints one JSON line per configuration."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/orb-slam2-chinesenotes_amd/pyhost")
import torch  # noqa: E402
from orbhip import capi, synth  # noqa: E402

MBF = 386.1448
MB = MBF / 718.856


def config3(n_iter=100):
    w, h, nf = 1241, 376, 2000
    lefts = [synth.synth_frame(100 + i, w, h) for i in range(4)]
    rights = [synth.synth_stereo_right(100 + i, w, h) for i in range(4)]
    exl, exr = capi.Extractor(nf), capi.Extractor(nf)
    ts, matched = [], 0
    for it in range(n_iter + 5):
        t0 = time.perf_counter()
        kl, dl = exl.extract(lefts[it % 4])
        kr, dr = exr.extract(rights[it % 4])
        u, z = capi.stereo_match(exl, exr, kl, dl, kr, dr, MB, MBF)
        dt = time.perf_counter() - t0
        if it >= 5:
            ts.append(dt * 1e3)
            matched += int((u >= 0).sum())
    ts = np.array(ts)
    return {"config": 3, "workload": "1241x376 stereo pair, nFeatures 2000, extract both + stereo search, host in / host out",
            "ms_per_pair_median": round(float(np.median(ts)), 3), "ms_per_pair_p90": round(float(np.percentile(ts, 90)), 3),
            "pairs_per_s": round(1e3 / float(np.median(ts)), 1), "mean_stereo_matches": round(matched / n_iter, 1)}


def config5(n_kf=1000, n_stream=200):
    W, H = 752, 480
    dev = torch.device("cuda", 0)
    ex, mt = capi.Extractor(), capi.Matcher(0.7, True)
    cap = ex.max_keypoints
    F = n_kf + 2                                                  # slots n_kf, n_kf+1 hold stream frames alternately
    d_kps = torch.zeros(F * cap * 28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(F * cap * 32, dtype=torch.uint8, device=dev)
    d_counts = torch.zeros(F, dtype=torch.int32, device=dev)
    d_node = torch.zeros(F * cap, dtype=torch.int16, device=dev)
    d_valid = torch.from_numpy(np.stack([synth.synth_valid_flags(cap, 7000 + i) for i in range(F)])).to(dev)
    d_cent = torch.from_numpy(synth.synth_vocabulary()).to(dev)
    # the keyframe DB: descriptors of the first n_kf frames (64 distinct images repeated keep the set-up short)
    base = torch.from_numpy(synth.synth_batch(0, 64, W, H)).to(dev)
    for k0 in range(0, n_kf, 64):
        n = min(64, n_kf - k0)
        ex.extract_batch_device(base.data_ptr(), n, H, W, W, W * H, d_kps.data_ptr() + k0 * cap * 28,
                                d_desc.data_ptr() + k0 * cap * 32, cap, d_counts.data_ptr() + k0 * 4)
        ex.sync()
    mt.bow_assign_device(d_desc.data_ptr(), d_counts.data_ptr(), n_kf, cap, d_cent.data_ptr(), d_node.data_ptr())
    mt.sync()
    stream = torch.from_numpy(synth.synth_batch(1000, 16, W, H)).to(dev)
    kf_idx = torch.arange(n_kf, dtype=torch.int32, device=dev)
    f_idx = [torch.full((n_kf,), n_kf + s, dtype=torch.int32, device=dev) for s in (0, 1)]
    d_match = [torch.zeros(n_kf * cap, dtype=torch.int32, device=dev) for _ in (0, 1)]
    d_nm = [torch.zeros(n_kf, dtype=torch.int32, device=dev) for _ in (0, 1)]
    store = dict(desc=d_desc.data_ptr(), kps=d_kps.data_ptr(), valid=d_valid.data_ptr(), counts=d_counts.data_ptr(),
                 node_of=d_node.data_ptr(), cap=cap, n_frames=F)
    torch.cuda.synchronize()

    def extract(i):                                # stream frame i -> query slot i % 2
        s = i % 2
        ex.extract_batch_device(stream.data_ptr() + (i % 16) * W * H, 1, H, W, W, W * H, d_kps.data_ptr() + (n_kf + s) * cap * 28,
                                d_desc.data_ptr() + (n_kf + s) * cap * 32, cap, d_counts.data_ptr() + (n_kf + s) * 4)

    def match(i):
        s = i % 2
        mt.bow_assign_device(d_desc.data_ptr() + (n_kf + s) * cap * 32, d_counts.data_ptr() + (n_kf + s) * 4, 1, cap, d_cent.data_ptr(),
                             d_node.data_ptr() + (n_kf + s) * cap * 2)
        mt.match_bow_batch_device(store, kf_idx.data_ptr(), f_idx[s].data_ptr(), n_kf, d_match[s].data_ptr(), d_nm[s].data_ptr())

    def run(n):
        # software pipeline over two query slots: match(i) runs beside extract(i+1).  Both stream waits are taken
        # BEFORE the two launches, so match(i) waits for extract(i) only and extract(i+1) for match(i-1) only (which
        # used the slot extract(i+1) is about to overwrite).
        ex.wait_for(mt.stream)
        extract(0)
        for i in range(n):
            mt.wait_for(ex.stream)
            ex.wait_for(mt.stream)
            extract(i + 1)
            match(i)

    run(5)
    ex.sync(); mt.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n_stream)
    ex.sync(); mt.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_stream
    d_nm = d_nm[0]
    return {"config": 5, "workload": "752x480 stream, per frame extract + SearchByBoW against a %d-keyframe DB in HBM" % n_kf,
            "ms_per_frame": round(dt * 1e3, 3), "frames_per_s": round(1.0 / dt, 1),
            "pair_matches_per_s": round(n_kf / dt, 0), "mean_matches_per_pair": round(float(d_nm.float().mean().item()), 2)}


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "5"]
    if "3" in which:
        print(json.dumps(config3()))
    if "5" in which:
        print(json.dumps(config5()))
