#!/usr/bin/env python3
"""bench.py -- frames/s of the MI355X-native ORB front-end (extract + match), one JSON line.

A "step" is one pass of the hot path over one batch of synthetic frames already resident in HBM:
    ORBextractor::operator() on B frames  (pyramid -> FAST cells -> quadtree -> orientation+rBRIEF)
  + vocabulary-node assignment (synthetic stand-in for DBoW2 transform)
  + ORBmatcher::SearchByBoW for B (keyframe, frame) pairs (frame i as keyframe vs frame i+1).
Workload at N=1 = BASELINE.json configs[1]/[3]: 640x480, 8 levels, 1000 features, batched
(configs[3] is the same frame shape sharded over GPUs; per-GPU batch is fixed -> weak scaling).

Multi-GPU: one process per GPU (torchrun), frames sharded by rank, NO data-path collective; the
only collective is the RCCL broadcast of the BRIEF pattern from rank 0 at start-up.

The CPU oracle is used ONLY in the cpu_baseline leg (rank 0, N=1), on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))

import numpy as np
import torch

from orbhip import capi, shard, synth

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes_per_frame(cols, rows, pyr_px, n_kp):
    # input read + pyramid written once + keypoint/descriptor records
    return cols * rows + pyr_px + n_kp * (28 + 32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--input-sets", type=int, default=2,
                    help="distinct input batches that consecutive steps alternate between: one 157 MB batch re-read every "
                         "step would sit in the 256 MB Infinity Cache and flatter the pyramid's first kernel")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="number of independent (extractor, matcher, output buffers) lanes that consecutive steps alternate "
                         "between; lanes run on their own streams, so step k+1 overlaps step k")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-gpu", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--no-match", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (CPU rehearsal of the N>1 path)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")      # where collective tensors live
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    B, W, H = args.frames_per_gpu, args.width, args.height
    ex = capi.Extractor(args.nfeatures, 1.2, 8, 20, 7, device=local_rank)
    mt = capi.Matcher(0.7, True, device=local_rank)           # Tracking.cc:815 parameters
    cap = ex.max_keypoints

    # ---- the one collective: rank 0 broadcasts the BRIEF pattern (RCCL over xGMI)
    pat = torch.zeros(1024, dtype=torch.int8, device=comm_dev)
    if rank == 0:
        pat.copy_(torch.from_numpy(capi.builtin_pattern()))
    shard.broadcast_pattern(dist, pat, 0)
    pat = pat.to(dev)
    torch.cuda.synchronize()
    ex.set_pattern_device(pat.data_ptr())

    # ---- synthetic inputs, resident in HBM before the timed region
    first, _ = shard.weak_range(B, rank)            # weak scaling: every rank owns B frames of its own
    n_sets = max(1, args.input_sets)
    frames_sets = [synth.synth_batch(first + k * world * B, B, W, H) for k in range(n_sets)]
    d_img_sets = [torch.from_numpy(fr).to(dev) for fr in frames_sets]
    d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    d_counts = torch.zeros(B, dtype=torch.int32, device=dev)
    d_nodeof = torch.zeros(B * cap, dtype=torch.int16, device=dev)
    d_valid = torch.from_numpy(np.stack([synth.synth_valid_flags(cap, rank * B + i) for i in range(B)])).to(dev)
    d_cent = torch.from_numpy(synth.synth_vocabulary()).to(dev)
    kf_idx = torch.arange(B, dtype=torch.int32, device=dev)
    f_idx = ((torch.arange(B, dtype=torch.int32, device=dev) + 1) % B).to(torch.int32)
    d_match = torch.zeros(B * cap, dtype=torch.int32, device=dev)
    d_nm = torch.zeros(B, dtype=torch.int32, device=dev)
    store = dict(desc=d_desc.data_ptr(), kps=d_kps.data_ptr(), valid=d_valid.data_ptr(), counts=d_counts.data_ptr(),
                 node_of=d_nodeof.data_ptr(), cap=cap, n_frames=B)
    torch.cuda.synchronize()

    # ---- lanes: lane 0 is (ex, mt, the buffers above); further lanes (--pipeline P) have their own extractor, matcher
    # and output buffers and share the read-only inputs.  Consecutive steps alternate between lanes, each lane on its
    # own streams, so the latency-bound stages of one step (pyramid, quadtree, matcher) overlap the issue-bound stages
    # (FAST, descriptors) of the next.
    lanes = [dict(ex=ex, mt=mt, kps=d_kps, desc=d_desc, counts=d_counts, nodeof=d_nodeof, match=d_match, nm=d_nm, store=store)]
    for _ in range(1, max(1, args.pipeline)):
        lex = capi.Extractor(args.nfeatures, 1.2, 8, 20, 7, device=local_rank)
        lex.set_pattern_device(pat.data_ptr())
        lmt = capi.Matcher(0.7, True, device=local_rank)
        ln = dict(ex=lex, mt=lmt, kps=torch.zeros_like(d_kps), desc=torch.zeros_like(d_desc), counts=torch.zeros_like(d_counts),
                  nodeof=torch.zeros_like(d_nodeof), match=torch.zeros_like(d_match), nm=torch.zeros_like(d_nm))
        ln["store"] = dict(desc=ln["desc"].data_ptr(), kps=ln["kps"].data_ptr(), valid=d_valid.data_ptr(),
                           counts=ln["counts"].data_ptr(), node_of=ln["nodeof"].data_ptr(), cap=cap, n_frames=B)
        lanes.append(ln)
    torch.cuda.synchronize()
    step_no = [0]

    def step():
        ln = lanes[step_no[0] % len(lanes)]
        ln["set"] = step_no[0] % n_sets                # which input batch this lane's buffers will hold results of
        d_imgs = d_img_sets[ln["set"]]
        step_no[0] += 1
        lx, lm = ln["ex"], ln["mt"]
        lx.wait_for(lm.stream)                     # this lane's outputs of its previous step are still being matched
        lx.extract_batch_device(d_imgs.data_ptr(), B, H, W, W, W * H, ln["kps"].data_ptr(), ln["desc"].data_ptr(), cap,
                                ln["counts"].data_ptr())
        if not args.no_match:
            lm.wait_for(lx.stream)
            lm.bow_assign_device(ln["desc"].data_ptr(), ln["counts"].data_ptr(), B, cap, d_cent.data_ptr(),
                                 ln["nodeof"].data_ptr())
            lm.match_bow_batch_device(ln["store"], kf_idx.data_ptr(), f_idx.data_ptr(), B, ln["match"].data_ptr(),
                                      ln["nm"].data_ptr())

    def full_sync():
        for ln in lanes:
            ln["ex"].sync()
            ln["mt"].sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    full_sync()
    ex.set_profiling(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    full_sync()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(dist, elapsed, comm_dev)
    frames_done = shard.sum_over_ranks(dist, B * args.steps, comm_dev)

    stage_ms = ex.stage_ms()                       # HIP events on the stream the kernels were launched on
    launch_frames = ex.profiled_frames()           # frames per timed launch (sub-batch 0)
    ex.set_profiling(False)
    counts = d_counts.cpu().numpy()
    nm = d_nm.cpu().numpy()
    mean_kp = float(counts.mean())
    pyr_px = sum(int(ex.pyramid_level(0, l).size) for l in range(8))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_frames = frames_done
    fps = total_frames / elapsed
    names = ["pyramid(k_copy_level0 + resize kernels)", "k_fast_cells", "k_quadtree", "k_orient_desc"]
    dom = int(np.argmax(stage_ms[:4]))
    bytes_frame = algorithmic_bytes_per_frame(W, H, pyr_px, mean_kp)
    achieved = bytes_frame * launch_frames / (float(stage_ms[dom]) * 1e-3) / 1e9
    out = {
        "metric": "frames/sec ORB extract+match, 640x480 8-level 1000-feat; HBM GB/s vs peak",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "batch of %d synthetic %dx%d frames per GPU, nFeatures=%d, 8 levels, scale 1.2, "
                               "FAST 20/7; extract + SearchByBoW(frame i as keyframe, frame i+1), ratio 0.7"
                               % (B, W, H, args.nfeatures),
                   "frames_per_gpu": B, "input_sets": n_sets, "match": not args.no_match, "mean_keypoints": round(mean_kp, 1),
                   "mean_bow_matches": round(float(nm.mean()), 1),
                   "frames_per_launch": launch_frames,
                   "stage_ms_per_launch": {n: round(float(v), 4) for n, v in zip(names + ["extract_total"], stage_ms)}},
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                     "algorithmic_bytes_per_frame": int(bytes_frame), "frames_per_launch": launch_frames,
                     "kernel_ms_per_launch": round(float(stage_ms[dom]), 4)},
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # filled by tools/pmc_summary.py from rocprofv3 --pmc runs
    if os.path.exists(pmc):
        try:
            tr = json.load(open(pmc))
            key = names[dom].split("(")[0]
            if tr.get("frames_per_launch") == launch_frames and key in tr.get("bytes_per_launch", {}):
                out["roofline"]["traffic"] = tr["bytes_per_launch"][key]
        except Exception:
            pass

    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, frames_sets[lanes[0]["set"]], d_kps, d_desc, counts, cap, nm, d_match)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(args, frames_np, d_kps, d_desc, counts, cap, nm_gpu, d_match):
    """The CPU oracle (kind 'port': the reference itself needs OpenCV/DBoW2 and cannot be built),
    single thread, on a bounded sample of the same frames; also the parity check of that sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    ref = oracle.Extractor(args.nfeatures, 1.2, 8, 20, 7)
    cent = synth.synth_vocabulary()
    B = frames_np.shape[0]
    kps_gpu = d_kps.cpu().numpy().view(capi.KP_DTYPE).reshape(B, cap)
    desc_gpu = d_desc.cpu().numpy().reshape(B, cap, 32)
    match_gpu = d_match.cpu().numpy().reshape(B, cap)
    t_start = time.perf_counter()
    done, parity = 0, True
    feats = []
    t_extract = t_match = 0.0
    while done < B and (time.perf_counter() - t_start) < args.cpu_seconds:
        t0 = time.perf_counter()
        k, d = ref.extract(frames_np[done])
        t_extract += time.perf_counter() - t0
        n = int(counts[done])
        parity &= (n == len(k)) and kps_gpu[done, :n].tobytes() == k.tobytes() and np.array_equal(desc_gpu[done, :n], d)
        feats.append((k, d))
        if not args.no_match and done >= 1:
            t0 = time.perf_counter()
            (ka, da), (kb, db) = feats[done - 1], feats[done]
            fva, fvb = oracle.bow_transform(da, cent), oracle.bow_transform(db, cent)
            valid = synth.synth_valid_flags(cap, done - 1)[:len(ka)]
            nmr, mr = oracle.search_by_bow(da, ka["angle"], valid, fva, db, kb["angle"], fvb, 0.7, True)
            t_match += time.perf_counter() - t0
            parity &= (nmr == int(nm_gpu[done - 1])) and np.array_equal(mr, match_gpu[done - 1, :len(kb)])
        done += 1
    total = t_extract + t_match
    return {"value": round(done / total, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the %d benchmark frames, oracle extract%s, single thread" %
                      (done, B, "" if args.no_match else " + bow_transform + SearchByBoW"),
            "extract_ms_per_frame": round(t_extract / done * 1e3, 3),
            "match_ms_per_frame": round(t_match / max(done - 1, 1) * 1e3, 3),
            "gpu_matches_oracle_on_sample": bool(parity)}


if __name__ == "__main__":
    main()
