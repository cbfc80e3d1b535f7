#!/usr/bin/env python3
"""bench.py -- frames/s of the MI355X-native ORB front-end (extract + match), one JSON line.

Default workload (BASELINE.json configs[1]/[3], `--config c4`): a "step" is one pass of the hot path over one batch of
synthetic frames already resident in HBM:
    ORBextractor::operator() on B frames  (pyramid -> FAST strips -> quadtree -> orientation+rBRIEF)
  + DBoW2-shaped vocabulary descent (seeded k=10, L=6 tree, transform(..., levelsup=4): 60 Hamming per feature, the
    amount of work Frame::ComputeBoW does, reference src/Frame.cc:425-433)
  + ORBmatcher::SearchByBoW for B (keyframe, frame) pairs (frame i as keyframe vs frame i+1 of a moving-camera
    sequence, so consecutive frames share most corners), ratio 0.7 as src/Tracking.cc:815.
`value` is measured with TWO lanes (extractor, matcher, output buffers each) that consecutive steps alternate
between, so the latency-bound stages of one step overlap the issue-bound stages of the next; the per-kernel durations
behind `roofline` come from a separate single-lane (non-overlapped) pass of the same run, outside the timed region.
Other configurations: `--config c3` (KITTI-sized stereo pairs: extract both + stereo search) and `--config c5`
(752x480 stream against a 1000-keyframe descriptor DB) print the same contract line for their workloads.

Multi-GPU: one process per GPU (torchrun, or `--gpus N` alone, which starts the ranks itself), frames sharded by rank,
NO data-path collective; the only collective is the RCCL broadcast of the BRIEF pattern from rank 0 at start-up.
`--scaling weak` (default) gives every rank B frames; `--scaling strong` splits ONE batch of B frames over the ranks
(BASELINE.json configs[3] as written).

The CPU oracle is used ONLY in the cpu_baseline leg (rank 0, N=1), on a bounded sample.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))

import numpy as np


# ------------------------------------------------------------------ CPU-only worker (all-cores leg of cpu_baseline)
def cpu_worker(first, count, width, height, nfeatures):
    """Runs in a child process WITHOUT torch / GPU: oracle extraction of `count` sequence frames, wall time printed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from orbhip import synth
    frames = synth.synth_sequence(first, count, width, height)
    ref = oracle.Extractor(nfeatures, 1.2, 8, 20, 7)
    ref.extract(frames[0])
    t0 = time.perf_counter()
    for f in frames:
        ref.extract(f)
    print(json.dumps({"frames": count, "seconds": time.perf_counter() - t0}))


if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
    cpu_worker(*[int(v) for v in sys.argv[2:7]])
    sys.exit(0)

import torch

from orbhip import capi, shard, synth

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
N_SIMD, MAX_CLOCK_GHZ = 1024, 2.4
VALU_ISSUE_PEAK = N_SIMD * MAX_CLOCK_GHZ / 2.0      # G wave-instructions/s: a wave64 VALU instruction issues over 2 cycles
MBF = 386.1448
MB = MBF / 718.856
METRIC = "frames/sec ORB extract+match, 640x480 8-level 1000-feat; HBM GB/s vs peak"
STAGES = ["pyramid(k_pyr_chain launches)", "k_fast_strips_p", "k_quadtree", "k_orient_desc"]


def algorithmic_bytes_per_frame(cols, rows, pyr_px, n_kp):
    # input read + pyramid written once + keypoint/descriptor records
    return cols * rows + pyr_px + n_kp * (28 + 32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: a timed region of ~0.35 s at 512 frames per step)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c4", choices=["c4", "c3", "c5"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--input-sets", type=int, default=2,
                    help="distinct input batches that consecutive steps alternate between: one 157 MB batch re-read every "
                         "step would sit in the 256 MB Infinity Cache and flatter the pyramid's first kernel")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="number of independent (extractor, matcher, output buffers) lanes that consecutive steps alternate "
                         "between; lanes run on their own streams, so step k+1 overlaps step k.  Default (0): 4 lanes for "
                         "batches of >= 256 frames per launch (measured 2 / 3 / 4 / 6 / 8 lanes: 319 / 320 / 327 / 329 / 328 k "
                         "frames/s), 2 below (64-frame batches: 276 / 273 / 265 k with 2 / 4 / 8 -- the host's launch rate)")
    ap.add_argument("--frames-per-gpu", type=int, default=512, help="batch size B (weak: per GPU; strong: in total)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--content", default="shapes", choices=["shapes", "natural"],
                    help="synthetic frame content: drawn rectangles + discs + noise (SURVEY 8d, default) or natural image statistics "
                         "(1/f texture, occluding objects, blur, illumination ramp: orbhip.synth.synth_natural)")
    ap.add_argument("--c3-two-handles", action="store_true",
                    help="config c3: left and right images through two extractor handles on two streams (the reference's two "
                         "ORBextractor objects, src/Frame.cc:82-85) instead of ONE 64-frame batch [L0..L31, R0..R31] through one "
                         "handle (the default since round 4: 207 k against 190 k frames/s)")
    ap.add_argument("--c5-pair-kernel", action="store_true", help="config 5: SearchByBoW with the pair kernel (one workgroup per (keyframe, frame) pair) instead of the query form")
    ap.add_argument("--c5-no-minibatch", action="store_true", help="config 5: skip the mini-batch variant (the counter passes: every search launch is one query)")
    ap.add_argument("--c5-bow-early", action="store_true", help="config 5: vocabulary descent + feature vector of a frame behind its extraction "
                    "(own handle per extractor) instead of in front of its search (measured slower)")
    ap.add_argument("--c5-extractors", type=int, default=2, help="config 5: extractor handles that consecutive stream frames alternate between")
    ap.add_argument("--c5-three-calls", action="store_true", help="config 5: ComputeBoW and the search as three calls of the C ABI "
                    "(descent, feature vector, search) instead of the fused orb_bow_query_frames_device")
    ap.add_argument("--c5-matchers", type=int, default=2, help="config 5: matcher handles that consecutive stream frames alternate between "
                    "(round 5: 2 -- ComputeBoW of frame i + 1 beside the search of frame i: 0.077 -> 0.064 ms per frame)")
    ap.add_argument("--c5-slots", type=int, default=6, help="config 5: query slots in the ring between the extractor and the matcher streams "
                    "(round 5: 6 with two matcher handles: 0.064 -> 0.051 ms per frame)")
    ap.add_argument("--c5-python-loop", action="store_true", help="config 5: time the frame loop as bench.py's Python loop submits it (the "
                    "interpreter is then what bounds the step); default: the same loop from C (tools/c5_loop.c through ctypes), "
                    "the Python figure reported beside it")
    ap.add_argument("--stream-frames", type=int, default=256,
                    help="config c5: distinct stream frames resident in HBM that the timed steps walk through (3682 = the whole "
                         "EuRoC MH01-sized sequence of BASELINE configs[4]; generating them on the host takes about a minute)")
    ap.add_argument("--no-match", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=200, help="timed frames of the single-thread cpu_baseline (+10 warm-up)")
    ap.add_argument("--no-cpu-all-cores", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure roofline.traffic live (two rocprofv3 --pmc child "
                    "runs of a 3-step bench); use the committed profiles/pmc_traffic.json instead")
    ap.add_argument("--no-natural", action="store_true", help="skip config.natural_content_fps (a child run of the same workload on "
                    "natural-statistics content)")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive host_in_host_out_fps measurement "
                    "(profiling runs: keeps every kernel launch at the benchmark's batch size)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (CPU rehearsal of the N>1 path)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--master-port", type=int, default=29511)
    return ap.parse_args()


def profile_path(name):
    """profiles/<name> for the counters bench.py falls back to when it cannot measure them live: the NEWEST committed round's
    artefact (profiles/rNN_b512_valu.json, rNN_b512_pmc_traffic.json, rNN_c5_pmc_traffic.json), never an unversioned copy."""
    import glob
    pat = {"pmc_traffic.json": "r[0-9][0-9]_b512_pmc_traffic.json", "valu.json": "r[0-9][0-9]_b512_valu.json",
           "c5_pmc_traffic.json": "r[0-9][0-9]_c5_pmc_traffic.json"}.get(name)
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", pat))) if pat else []
    return cands[-1] if cands else os.path.join(ROOT, "profiles", name)


_PAD_STREAMS = []


def hip_pad_streams(n):
    """Experiment knob (ORB_BENCH_STREAM_PADS): n idle HIP streams created -- and kept -- at this point.  HIP spreads streams over
    4 hardware queues by load; streams that share a queue serialise, so WHICH of a configuration's streams share one decides
    how its kernels overlap.  Idle pads shift where the streams created after them land."""
    if n <= 0:
        return
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    for _ in range(n):
        st = C.c_void_p()
        if hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0:
            _PAD_STREAMS.append(st)


def load_c5_loop():
    """tools/libc5loop.so (tools/c5_loop.c: the frame loop of config 5 from C), built on demand; None when it cannot be built."""
    import ctypes as C
    so = os.path.join(ROOT, "tools", "libc5loop.so")
    src = os.path.join(ROOT, "tools", "c5_loop.c")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "orb-slam2-chinesenotes_amd"), "c5-loop"], check=True, capture_output=True, timeout=120)
        lib = C.CDLL(so)
        lib.c5_loop_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        lib.c5_loop_run.restype = C.c_int
        return lib
    except Exception:
        return None


def load_profile(name):
    try:
        return json.load(open(profile_path(name)))
    except Exception:
        return None


def live_counters(kernel, launch_frames, content="shapes", child_args=None):
    """Counters of `kernel` per launch, measured NOW: three child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE and
    WRITE_SIZE need separate passes, the SQ counters a third; counter passes carry --kernel-trace only), 3 steps each,
    parsed like tools/pmc_summary.py (full-batch launches only; FETCH x2, profiles/r02_fetch_calibration.json).
    Returns {"traffic": bytes, "valu": {...}} with None for what could not be measured."""
    import csv
    import glob
    import shutil
    import tempfile
    res = {"traffic": None, "valu": None}
    if shutil.which("rocprofv3") is None:
        return res
    # already running under a profiler (the tool preloads itself into children): do not nest, use the committed profile
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return res
    tmp = tempfile.mkdtemp(prefix="orb_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    name = lambda r: r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
    grid = lambda r: int(r["Grid_Size"])
    passes = {"FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"],
              "SQ": ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT",
                     "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]}
    vals = {}
    try:
        for tag, ctrs in passes.items():
            out = os.path.join(tmp, tag)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", out, "-o", "run", "--",
                   sys.executable, os.path.abspath(__file__)]
            cmd += child_args if child_args is not None else ["--steps", "3", "--warmup", "1", "--pipeline", "1", "--no-cpu-baseline",
                                                              "--no-host-path", "--no-live-traffic", "--no-natural", "--frames-per-gpu", str(launch_frames),
                                                              "--content", content]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                continue
            rows, trace = [], []
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                rows += list(csv.DictReader(open(f)))
            for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
                trace += list(csv.DictReader(open(f)))
            mine = [r for r in rows if name(r) == kernel]
            if not mine:
                continue
            big = max(grid(r) for r in mine)                       # the benchmark's launches, not the vocabulary-training batch
            for c in ctrs:
                sel = [float(r["Counter_Value"]) for r in mine if grid(r) == big and r["Counter_Name"] == c]
                if sel:
                    vals[c] = sum(sel) / len(sel)
            if tag == "SQ":
                ids = {r["Dispatch_Id"] for r in mine if grid(r) == big}
                d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace if r.get("Dispatch_Id") in ids]
                if d:
                    vals["duration_ns_sq_pass"] = sum(d) / len(d)
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            res["traffic"] = int(2.0 * vals["FETCH_SIZE"] * 1024.0 + vals["WRITE_SIZE"] * 1024.0)
        if "SQ_INSTS_VALU" in vals and vals.get("SQ_WAVES"):
            act = vals.get("SQ_ACTIVE_INST_VALU", 0.0)
            cyc = vals.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            res["valu"] = {"valu_insts": vals["SQ_INSTS_VALU"], "waves": vals["SQ_WAVES"],
                           "valu_insts_per_wave": round(vals["SQ_INSTS_VALU"] / vals["SQ_WAVES"], 1),
                           "valu_busy": round(act * 4 / (N_SIMD * cyc), 4) if cyc else None,
                           "cycles_per_valu_inst": round(act * 4 / vals["SQ_INSTS_VALU"], 2),
                           "active_lane_frac": round(vals.get("SQ_THREAD_CYCLES_VALU", 0.0) / (act * 64), 4) if act else None,
                           "lds_conflict_frac": round(vals.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(vals.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4),
                           "kernel_us_in_counter_pass": round(vals.get("duration_ns_sq_pass", 0.0) / 1e3, 2)}
        return res
    except Exception:
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline_blocks(stage_ms, launch_frames, bytes_frame, extract_total_ms, live=False, content="shapes", child_args=None):
    """The contract `roofline` block (HBM, as SURVEY 8(d) defines `achieved`) for the dominant extractor kernel, plus
    `roofline_valu`: what actually binds FAST and the descriptor kernel is vector-instruction issue.  With live=True the
    kernel's HBM-side traffic AND its vector-instruction counters are measured in this run (three rocprofv3 child passes);
    otherwise (and as a fallback) they come from the committed profiles/pmc_traffic.json / profiles/valu.json."""
    dom = int(np.argmax(stage_ms[:4]))
    kern_s = float(stage_ms[dom]) * 1e-3
    achieved = bytes_frame * launch_frames / kern_s / 1e9
    rf = {"bound": "hbm", "kernel": STAGES[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_frame": int(bytes_frame),
          "frames_per_launch": launch_frames, "kernel_ms_per_launch": round(float(stage_ms[dom]), 4),
          "pipeline_achieved_GBs": round(bytes_frame * launch_frames / (extract_total_ms * 1e-3) / 1e9, 2),
          "binding": ("valu_issue (see roofline_valu): the kernel moves ~1 MB/frame from L2, HBM is not what limits it"
                      if STAGES[dom].startswith(("k_fast", "k_orient")) else
                      "vector-instruction issue of the fixed-point resampling inside barrier-separated band steps (~50 % VALU-busy), not HBM")}
    key = STAGES[dom].split("(")[0]
    tr = load_profile("pmc_traffic.json")
    lc = live_counters(key, launch_frames, content, child_args) if live else {"traffic": None, "valu": None}
    if lc["traffic"] is not None:
        rf["traffic"] = lc["traffic"]
        rf["traffic_source"] = "measured in this run: child runs of bench.py (3 steps) under rocprofv3 --pmc FETCH_SIZE / " \
                               "--pmc WRITE_SIZE, per full-batch launch; FETCH x2 (gfx950: 128-byte requests tallied at 64 B, " \
                               "profiles/r02_fetch_calibration.json)"
        if tr and key in tr.get("bytes_per_launch", {}):
            rf["traffic_committed_profile"] = int(tr["bytes_per_launch"][key] * launch_frames / tr["frames_per_launch"])
    elif tr and key in tr.get("bytes_per_launch", {}):
        rf["traffic"] = int(tr["bytes_per_launch"][key] * launch_frames / tr["frames_per_launch"])
        rf["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2 per " \
                               "profiles/r02_fetch_calibration.json), scaled to this launch size" % os.path.basename(profile_path("pmc_traffic.json"))
    rv = None
    k, src = lc["valu"], "measured in this run: a child run of bench.py (3 steps) under rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ..."
    if k is None:
        vj = load_profile("valu.json")
        if vj and key in vj.get("kernels", {}):
            k = dict(vj["kernels"][key])
            k["valu_insts"] = k["valu_insts"] * launch_frames / vj["frames_per_launch"]
            src = "profiles/%s (tools/prof_collect.sh + tools/pmc_summary.py), scaled to this launch size" % os.path.basename(profile_path("valu.json"))
    if k is not None:
        insts = k["valu_insts"]
        ach = insts / kern_s / 1e9
        rv = {"bound": "valu_issue", "kernel": key, "achieved": round(ach, 2), "peak": VALU_ISSUE_PEAK, "unit": "G wave-inst/s",
              "frac": round(ach / VALU_ISSUE_PEAK, 4), "valu_insts_per_launch": int(insts),
              "valu_insts_per_wave": k.get("valu_insts_per_wave"), "valu_busy": k.get("valu_busy"),
              "cycles_per_valu_inst": k.get("cycles_per_valu_inst"), "active_lane_frac": k.get("active_lane_frac"),
              "lds_conflict_frac": k.get("lds_conflict_frac"), "source": src,
              "note": "peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction; this kernel's mix (v_perm_b32, "
                      "v_pk_*3_f16) averages cycles_per_valu_inst cycles, so frac x that / 2 is the share of issue cycles used"}
    return rf, rv


def trained_vocabulary(ex, W, H, n_train=8):
    """synth.synth_vocab_tree_trained on the descriptors the GPU extractor finds in n_train sequence frames (the same
    frames on every rank, bit-exact results, so every rank builds the same tree)."""
    sample = np.concatenate([d for _, d in ex.extract_batch(synth.synth_sequence(4000, n_train, W, H))])
    return synth.synth_vocab_tree_trained(sample, 10, 6)


# ================================================================== config c4 (default)
def phase_a_survivor_rate(ex, low_th):
    """Share of pixel PAIRS of the detection zones that survive the FAST kernel's cheap rejection (phase A: the bound U of
    csrc/orb_fast.hip over the four even ring pairs exceeds min(iniTh, minTh) in one of the two pixels), evaluated in numpy
    on the device-resident pyramid of frame 0 -- a statistic of the image content, reported next to the throughput."""
    tot = sur = 0
    for l in range(8):
        I = ex.pyramid_level(0, l).astype(np.int16)
        h, w = I.shape
        if h < 40 or w < 40:
            continue
        c = I[19:h - 19, 19:w - 19]
        ring = lambda dx, dy: I[19 + dy:h - 19 + dy, 19 + dx:w - 19 + dx]
        pairs = [(ring(0, 3), ring(0, -3)), (ring(3, 0), ring(-3, 0)), (ring(2, 2), ring(-2, -2)), (ring(2, -2), ring(-2, 2))]
        mlo = np.maximum.reduce([np.minimum(a, b) for a, b in pairs])
        mhi = np.minimum.reduce([np.maximum(a, b) for a, b in pairs])
        u = np.maximum(c - mlo, mhi - c) > low_th
        ww = u.shape[1] & ~1
        pr = u[:, 0:ww:2] | u[:, 1:ww:2]
        tot += pr.size
        sur += int(pr.sum())
    return round(sur / max(tot, 1), 4)


def run_c4(args, rank, local_rank, world, dev, comm_dev, dist):
    W, H = args.width, args.height
    seq = (lambda first, n: synth.synth_sequence(first, n, W, H, content=args.content))
    ex0 = capi.Extractor(args.nfeatures, 1.2, 8, 20, 7, device=local_rank)
    cap = ex0.max_keypoints

    # ---- the one collective: rank 0 broadcasts the BRIEF pattern (RCCL over xGMI)
    pat = torch.zeros(1024, dtype=torch.int8, device=comm_dev)
    if rank == 0:
        pat.copy_(torch.from_numpy(capi.builtin_pattern()))
    shard.broadcast_pattern(dist, pat, 0)
    pat = pat.to(dev)
    torch.cuda.synchronize()
    ex0.set_pattern_device(pat.data_ptr())

    # ---- vocabulary: complete k=10, L=6 tree (ORBvoc is absent) whose top levels are trained, as DBoW2 trains its
    # vocabularies, on the descriptors of 8 benchmark frames (extracted here, by the product path, before any timing)
    sample = np.concatenate([d for _, d in ex0.extract_batch(seq(4000, 8))])
    tree = synth.synth_vocab_tree_trained(sample, 10, 6)
    voc = capi.Vocabulary(tree, device=local_rank)
    n_nodes = voc.level_nodes(4)

    def one_pass(scaling, first_ex):
        """The whole measurement for one scaling mode: `weak` = every rank its own batch of --frames-per-gpu frames, `strong` =
        ONE batch of --frames-per-gpu frames split over the ranks (BASELINE configs[3] as written)."""
        if scaling == "strong":
            first, B = shard.frame_range(args.frames_per_gpu, world, rank)
        else:
            first, B = shard.weak_range(args.frames_per_gpu, rank)
        if B <= 0:
            raise SystemExit("rank %d has no frames" % rank)
        n_lanes = args.pipeline if args.pipeline > 0 else (4 if B >= 256 else 2)
        # ---- synthetic inputs, resident in HBM before the timed region
        n_sets = max(1, args.input_sets)
        set_stride = (args.frames_per_gpu * world + 7) // 8 * 8      # whole scenes between the input sets
        frames_sets = [seq(first + k * set_stride, B) for k in range(n_sets)]
        d_img_sets = [torch.from_numpy(fr).to(dev) for fr in frames_sets]
        valid_np = np.stack([synth.synth_valid_flags(cap, first + i) for i in range(B)])
        d_valid = torch.from_numpy(valid_np).to(dev)
        kf_idx = torch.arange(B, dtype=torch.int32, device=dev)
        f_idx = ((torch.arange(B, dtype=torch.int32, device=dev) + 1) % B).to(torch.int32)

        def new_lane(lex, lmt):
            ln = dict(ex=lex, mt=lmt, kps=torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev),
                      desc=torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev), counts=torch.zeros(B, dtype=torch.int32, device=dev),
                      nodeof=torch.zeros(B * cap, dtype=torch.int16, device=dev), match=torch.zeros(B * cap, dtype=torch.int32, device=dev),
                      nm=torch.zeros(B, dtype=torch.int32, device=dev), set=0,
                      ckeys=torch.zeros(B * cap, dtype=torch.int32, device=dev), cstart=torch.zeros(B * n_nodes, dtype=torch.int16, device=dev),
                      ccnt=torch.zeros(B * n_nodes, dtype=torch.int16, device=dev), words=torch.zeros(B * cap, dtype=torch.int32, device=dev))
            ln["store"] = dict(desc=ln["desc"].data_ptr(), kps=ln["kps"].data_ptr(), valid=d_valid.data_ptr(),
                               counts=ln["counts"].data_ptr(), node_of=ln["nodeof"].data_ptr(), cap=cap, n_frames=B, n_nodes=n_nodes,
                               csr_keys=ln["ckeys"].data_ptr(), csr_start=ln["cstart"].data_ptr(), csr_cnt=ln["ccnt"].data_ptr())
            return ln

        # lanes: consecutive steps alternate between lanes, each lane on its own streams, so the latency-bound stages of one
        # step (pyramid, quadtree, matcher) overlap the issue-bound stages (FAST, descriptors) of the next.
        lanes = [new_lane(first_ex, capi.Matcher(0.7, True, device=local_rank))]           # Tracking.cc:815 parameters
        for _ in range(1, n_lanes):
            lex = capi.Extractor(args.nfeatures, 1.2, 8, 20, 7, device=local_rank)
            lex.set_pattern_device(pat.data_ptr())
            lanes.append(new_lane(lex, capi.Matcher(0.7, True, device=local_rank)))
        torch.cuda.synchronize()
        step_no = [0]

        def step(only_lane=None):
            ln = lanes[step_no[0] % len(lanes)] if only_lane is None else lanes[only_lane]
            ln["set"] = step_no[0] % n_sets                # which input batch this lane's buffers will hold results of
            d_imgs = d_img_sets[ln["set"]]
            step_no[0] += 1
            lx, lm = ln["ex"], ln["mt"]
            lx.wait_for(lm.stream)                     # this lane's outputs of its previous step are still being matched
            lx.extract_batch_device(d_imgs.data_ptr(), B, H, W, W, W * H, ln["kps"].data_ptr(), ln["desc"].data_ptr(), cap,
                                    ln["counts"].data_ptr())
            if not args.no_match:
                lm.wait_for(lx.stream)
                # word id (the BowVector's key) and level-(L-4) node (the FeatureVector's key) of every feature: the full descent
                voc.transform_device(lm, ln["desc"].data_ptr(), ln["counts"].data_ptr(), B, cap, 4, d_word_of=ln["words"].data_ptr(),
                                     d_node_of=ln["nodeof"].data_ptr())
                # the FeatureVector of every frame once (Frame::ComputeBoW), not once per pair inside the matcher
                lm.build_csr_device(ln["nodeof"].data_ptr(), ln["counts"].data_ptr(), B, cap, n_nodes, ln["ckeys"].data_ptr(),
                                    ln["cstart"].data_ptr(), ln["ccnt"].data_ptr())
                lm.match_bow_batch_device(ln["store"], kf_idx.data_ptr(), f_idx.data_ptr(), B, ln["match"].data_ptr(),
                                          ln["nm"].data_ptr())

        def full_sync():
            for ln in lanes:
                ln["ex"].sync()
                ln["mt"].sync()
            torch.cuda.synchronize()

        # ---- preparation, before the W warm-up steps and whatever W is: three SYNCHRONISED steps on every lane, as a caller
        # that reads its results does -- every lane's self-tuning (FAST strip lengths, the quadtree's LDS sort capacity)
        # has settled and every lane has run on both input sets before anything is timed (with W = 5 and four
        # lanes, three lanes had seen one step)
        for k in range(3):
            for li in range(len(lanes)):
                step(li)
                full_sync()
        overflowed = None
        try:
            overflowed = [int(v) for v in lanes[0]["ex"].fast_overflows()[0]]
        except Exception:
            pass
        step_no[0] = 0
        for i in range(args.warmup):
            step()
        full_sync()
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
        return dict(scaling=scaling, first=first, B=B, lanes=lanes, step=step, full_sync=full_sync, elapsed=elapsed,
                    frames_done=frames_done, frames_sets=frames_sets, valid_np=valid_np, n_sets=n_sets, overflowed=overflowed)

    P = one_pass(args.scaling, ex0)
    other = None
    if world > 1:
        # one driver run yields both figures BASELINE configs[3] asks about: the other scaling mode, same contract (barrier,
        # max over ranks, whole-job frames), reported under config.other_scaling
        ex1 = capi.Extractor(args.nfeatures, 1.2, 8, 20, 7, device=local_rank)
        ex1.set_pattern_device(pat.data_ptr())
        Q = one_pass("strong" if args.scaling == "weak" else "weak", ex1)
        other = {"scaling": Q["scaling"], "value": round(Q["frames_done"] / Q["elapsed"], 2), "unit": "frames/s",
                 "ms_per_step": round(Q["elapsed"] / args.steps * 1e3, 4), "frames_per_gpu": Q["B"],
                 "frames_per_step_all_gpus": int(Q["frames_done"] // args.steps), "steps": args.steps,
                 "note": "same timed contract as `value` (barrier + synchronize on both sides, max over ranks, frames of all ranks)"}
        del Q
    B, lanes, step, full_sync, elapsed, frames_done = P["B"], P["lanes"], P["step"], P["full_sync"], P["elapsed"], P["frames_done"]
    frames_sets, valid_np, n_sets, first = P["frames_sets"], P["valid_np"], P["n_sets"], P["first"]
    ex = lanes[0]["ex"]
    # the same K steps once more, untimed for `value`: a clock ramp or a cold start inside the timed region would show
    # as a difference between the two
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    full_sync()
    repeat_ms = (time.perf_counter() - t0) / args.steps * 1e3

    # ---- single-lane pass, outside the timed region: per-kernel durations that describe each kernel running alone
    # (HIP events on the extractor's own stream)
    n_prof = 6
    step(0); full_sync()
    ex.set_profiling(True)
    t1 = time.perf_counter()
    for _ in range(n_prof):
        step(0)
    full_sync()
    single_ms = (time.perf_counter() - t1) / n_prof * 1e3
    stage_ms = ex.stage_ms()
    launch_frames = ex.profiled_frames()
    ex.set_profiling(False)
    pyr_px = sum(int(ex.pyramid_level(0, l).size) for l in range(8))
    # ---- what the content looks like to the kernels (statistics, not timings)
    content_stats = None
    if rank == 0:
        cands = np.stack([ex.level_counts(i)[1] for i in range(min(4, B))]).mean(axis=0)
        content_stats = {"fast_candidates_per_level": [round(float(v), 1) for v in cands],
                         "fast_strips_overflowed_per_level_in_a_settled_batch": P["overflowed"],
                         "phase_a_surviving_pair_rate": phase_a_survivor_rate(ex, 7)}
    # ---- PCIe-inclusive host path (never `value`): the same B frames from HOST memory through orb_extract_batch
    # (chunked H2D | kernel chain | D2H pipeline), once with pinned and once with pageable caller buffers; extract only
    host_fps = None
    if rank == 0 and world == 1 and not args.no_host_path:
        host_fps = {}
        fr = frames_sets[0]
        for kind in ("pinned", "pageable"):
            try:
                if kind == "pinned":
                    h_img = torch.from_numpy(fr).pin_memory()
                    h_kps = torch.zeros((B, cap, 28), dtype=torch.uint8).pin_memory()
                    h_desc = torch.zeros((B, cap, 32), dtype=torch.uint8).pin_memory()
                    a_img, a_kps, a_desc = h_img.numpy(), h_kps.numpy(), h_desc.numpy()
                else:
                    a_img, a_kps, a_desc = fr, np.zeros((B, cap, 28), np.uint8), np.zeros((B, cap, 32), np.uint8)
                a_cnt = np.zeros(B, np.int32)
                best = 1e9
                for _ in range(3):
                    t2 = time.perf_counter()
                    ex.extract_batch_into(a_img, a_kps, a_desc, a_cnt)
                    best = min(best, time.perf_counter() - t2)
                host_fps[kind] = round(B / best, 1)
            except Exception as e:                              # reported, never fatal for the contract line
                host_fps[kind] = "failed: %s" % e
        step(0); full_sync()                                   # lane 0 holds a device-path result again (the parity sample below)
    ln0 = lanes[0]
    counts = ln0["counts"].cpu().numpy()
    nm = ln0["nm"].cpu().numpy()
    mean_kp = float(counts.mean())

    if rank != 0:
        return None
    fps = frames_done / elapsed
    bytes_frame = algorithmic_bytes_per_frame(W, H, pyr_px, mean_kp)
    rf, rv = roofline_blocks(stage_ms, launch_frames, bytes_frame, float(stage_ms[4]), live=(world == 1 and not args.no_live_traffic),
                             content=args.content)
    out = {
        "metric": METRIC, "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "batch of %d synthetic %dx%d frames %s, nFeatures=%d, 8 levels, scale 1.2, FAST 20/7; extract + "
                               "vocabulary descent (k=10, L=6, levelsup 4) + SearchByBoW(frame i as keyframe, frame i+1), ratio 0.7"
                               % (args.frames_per_gpu, W, H, "per GPU" if args.scaling == "weak" else "in total, split over the GPUs",
                                  args.nfeatures),
                   "content": ("drawn rectangles + discs + uniform noise (SURVEY 8d)" if args.content == "shapes" else
                               "natural image statistics: 1/f value-noise texture, occluding objects, blur, illumination ramp, sensor noise "
                               "(orbhip.synth.synth_natural)"),
                   "frames_per_gpu": B, "input_sets": n_sets, "lanes": len(lanes), "match": not args.no_match,
                   "mean_keypoints": round(mean_kp, 1), "mean_bow_matches": round(float(nm.mean()), 1),
                   "content_stats": content_stats,
                   "vocabulary": "complete k=10 L=6 tree, top two levels k-majority-trained on 8 frames, %d nodes, %d level-(L-4) nodes" % (tree["node_desc"].shape[0], n_nodes),
                   "frames_per_launch": launch_frames,
                   "prepared": "3 synchronised steps on every lane before the warm-up steps",
                   "repeat_ms_per_step": round(repeat_ms, 4),
                   "single_lane_ms_per_step": round(single_ms, 4),
                   "single_lane_frames_per_s": round(B / single_ms * 1e3, 1),
                   "stage_ms_per_launch_single_lane": {n: round(float(v), 4) for n, v in zip(STAGES + ["extract_total"], stage_ms)},
                   "extract_only_frames_per_s_single_lane": round(launch_frames / float(stage_ms[4]) * 1e3, 1),
                   "transform_plus_match_ms_single_lane": round(single_ms - float(stage_ms[4]), 4),
                   "host_in_host_out_fps": host_fps},
        "roofline": rf,
    }
    if other:
        out["config"]["other_scaling"] = other
    if rv:
        out["roofline_valu"] = rv
    # the same workload on natural-statistics content (what TUM / KITTI / EuRoC frames look like to the FAST kernel: more pixels
    # survive its cheap rejection than on the drawn shapes), measured by a child run outside this run's timed region
    if world == 1 and args.content == "shapes" and not args.no_natural:
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--content", "natural", "--no-natural", "--no-cpu-baseline", "--no-host-path",
                   "--no-live-traffic", "--steps", str(max(10, min(args.steps, 60))), "--warmup", "5", "--frames-per-gpu", str(args.frames_per_gpu),
                   "--width", str(W), "--height", str(H), "--nfeatures", str(args.nfeatures)]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and line:
                nat = json.loads(line[-1])
                out["config"]["natural_content_fps"] = nat["value"]
                out["config"]["natural_content_stage_ms_single_lane"] = nat["config"]["stage_ms_per_launch_single_lane"]
        except Exception:
            pass
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline_c4(args, frames_sets[ln0["set"]], first, ln0, counts, cap, nm, tree, valid_np)
        cb["gpu_over_cpu_port"] = {"extract_plus_match": round(fps / cb["value"], 1),
                                   "extract_only": round(launch_frames / float(stage_ms[4]) * 1e3 / cb["extract_only_frames_per_s"], 1),
                                   "note": "against this single-thread scalar port (kind 'port'), not OpenCV's SIMD build of the reference"}
        out["cpu_baseline"] = cb
    return out


def pin_one_core():
    try:
        cores = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, {cores[len(cores) // 2]})
        return cores
    except Exception:
        return None


def cpu_baseline_c4(args, frames_np, first, ln0, counts, cap, nm_gpu, tree, valid_np):
    """The CPU oracle (kind 'port': the reference itself needs OpenCV/DBoW2 and cannot be built), as BASELINE.md plans it:
    ONE pinned thread, 10 warm-up frames then >= 200 timed frames, median per-frame time (extract and match timed
    separately); plus an all-cores figure (one frame stream per core, cores stated).  Every sampled frame is also the
    parity check of the GPU results of the timed run."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    ref = oracle.Extractor(args.nfeatures, 1.2, 8, 20, 7)
    B = frames_np.shape[0]
    kps_gpu = ln0["kps"].cpu().numpy().view(capi.KP_DTYPE).reshape(B, cap)
    desc_gpu = ln0["desc"].cpu().numpy().reshape(B, cap, 32)
    match_gpu = ln0["match"].cpu().numpy().reshape(B, cap)
    all_cores = pin_one_core()
    n_warm = min(10, max(B - 2, 0))
    n_timed = max(1, min(args.cpu_frames, B - n_warm))
    t_ext, t_mat, parity, feats = [], [], True, {}
    for i in range(n_warm + n_timed):
        t0 = time.perf_counter()
        k, d = ref.extract(frames_np[i])
        te = time.perf_counter() - t0
        n = int(counts[i])
        parity &= (n == len(k)) and kps_gpu[i, :n].tobytes() == k.tobytes() and np.array_equal(desc_gpu[i, :n], d)
        tm = 0.0
        fv = None
        if not args.no_match:
            t0 = time.perf_counter()
            _, nid = oracle.vocab_transform(tree, d, 4)
            fv = oracle.featvec_from_nodes(nid)
            tm += time.perf_counter() - t0
        feats[i] = (k, d, fv)
        if not args.no_match and i >= 1:
            (ka, da, fva), (kb, db, fvb) = feats[i - 1], feats[i]
            t0 = time.perf_counter()
            nmr, mr = oracle.search_by_bow(da, ka["angle"], valid_np[i - 1][:len(ka)], fva, db, kb["angle"], fvb, 0.7, True)
            tm += time.perf_counter() - t0
            parity &= (nmr == int(nm_gpu[i - 1])) and np.array_equal(mr, match_gpu[i - 1, :len(kb)])
            feats.pop(i - 1)
        if i >= n_warm:
            t_ext.append(te)
            t_mat.append(tm)
    if all_cores:
        os.sched_setaffinity(0, set(all_cores))
    med_e, med_m = statistics.median(t_ext), statistics.median(t_mat)
    res = {"value": round(1.0 / (med_e + med_m), 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "frames %d..%d of the benchmark batch after %d warm-up frames: oracle extract%s, one pinned thread, median "
                     "per-frame time" % (n_warm, n_warm + n_timed - 1, n_warm,
                                         "" if args.no_match else " + vocabulary descent + SearchByBoW"),
           "timed_frames": n_timed, "extract_ms_per_frame_median": round(med_e * 1e3, 3),
           "match_ms_per_frame_median": round(med_m * 1e3, 3),
           "extract_only_frames_per_s": round(1.0 / med_e, 3),
           "gpu_matches_oracle_on_sample": bool(parity)}
    # north_star asks for >= 1000x the single-thread CPU ORBextractor: the ratios against THIS port (not OpenCV's SIMD paths)
    res["gpu_over_cpu_port"] = {"extract_plus_match": None, "note": "filled by the caller"}
    if not args.no_cpu_all_cores and all_cores and len(all_cores) > 1:
        nproc = len(all_cores)
        per = 24
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(first + 8 * w), str(per),
                                   str(args.width), str(args.height), str(args.nfeatures)], stdout=subprocess.PIPE, text=True)
                 for w in range(nproc)]
        rate = 0.0
        for p in procs:
            o, _ = p.communicate(timeout=600)
            try:
                r = json.loads(o.strip().splitlines()[-1])
                rate += r["frames"] / r["seconds"]
            except Exception:
                pass
        res["all_cores"] = {"value": round(rate, 2), "unit": "frames/s (extract only)", "cores": nproc,
                            "sample": "%d concurrent single-thread oracle processes x %d frames each" % (nproc, per)}
    return res


# ================================================================== config c3: stereo pairs
def run_c3(args, rank, local_rank, world, dev, comm_dev, dist):
    """KITTI-sized stereo: a step = S pairs resident in HBM: extract S left + S right images (two handles on two
    streams, as the reference's two threads, src/Frame.cc:82-85) + ComputeStereoMatches (src/Frame.cc:513-699) per pair
    on the device-resident pyramids."""
    W, H, nf, S = 1241, 376, 2000, 32
    base = rank * S
    lefts = np.stack([synth.synth_frame(100 + base + i, W, H) for i in range(S)])
    rights = np.stack([synth.synth_stereo_right(100 + base + i, W, H) for i in range(S)])
    d_l, d_r = torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev)
    buf = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)
    one = not args.c3_two_handles
    d_lr = torch.cat([d_l, d_r]) if one else None
    # Lanes: independent (left handle, right handle, output buffers) sets that consecutive steps alternate between, as in
    # config c4.  Inside a lane everything is ordered (the stereo search of a step reads the right handle's pyramid and
    # keypoints, so the library orders the right handle's next extraction behind it); the search of one lane runs beside
    # the extractions of the other.
    # (two handles per lane, measured 1 / 2 / 3 / 4 / 6 lanes: 174 / 181 / 192 / 187 / 189 k frames/s; one handle per lane, round 4:
    # 2 / 3 / 4 lanes 207 / 201 / 190 k)
    n_lanes = args.pipeline if args.pipeline > 0 else (2 if one else 3)

    def make_lane():
        ln = {"exl": capi.Extractor(nf, device=local_rank)}
        ln["exr"] = ln["exl"] if one else capi.Extractor(nf, device=local_rank)
        cap_ = ln["exl"].max_keypoints
        if one:       # ONE batch [L0 .. L(S-1), R0 .. R(S-1)] through one handle; pair p = frames (p, S + p) of that batch
            ln["k2"], ln["d2"], ln["c2"] = buf(2 * S * cap_ * 28, torch.uint8), buf(2 * S * cap_ * 32, torch.uint8), buf(2 * S, torch.int32)
            ln["kl"], ln["kr"] = ln["k2"][:S * cap_ * 28], ln["k2"][S * cap_ * 28:]
            ln["dl"], ln["dr"] = ln["d2"][:S * cap_ * 32], ln["d2"][S * cap_ * 32:]
            ln["cl"], ln["cr"] = ln["c2"][:S], ln["c2"][S:]
        else:
            ln["kl"], ln["kr"] = buf(S * cap_ * 28, torch.uint8), buf(S * cap_ * 28, torch.uint8)
            ln["dl"], ln["dr"] = buf(S * cap_ * 32, torch.uint8), buf(S * cap_ * 32, torch.uint8)
            ln["cl"], ln["cr"] = buf(S, torch.int32), buf(S, torch.int32)
        ln["ur"], ln["dp"] = buf(S * cap_, torch.float32), buf(S * cap_, torch.float32)
        return ln

    lanes = [make_lane() for _ in range(n_lanes)]
    exl, exr = lanes[0]["exl"], lanes[0]["exr"]
    cap = exl.max_keypoints
    kl, dl, cl, cr, ur, dp = (lanes[0][k] for k in ("kl", "dl", "cl", "cr", "ur", "dp"))
    torch.cuda.synchronize()

    def step(i=0):
        ln = lanes[i % n_lanes]
        p = lambda k: ln[k].data_ptr()
        if one:
            ln["exl"].extract_batch_device(d_lr.data_ptr(), 2 * S, H, W, W, W * H, p("k2"), p("d2"), cap, p("c2"))
            capi.stereo_match_batch_device(ln["exl"], ln["exl"], 0, S, S, p("kl"), p("dl"), p("cl"), p("kr"), p("dr"), p("cr"), cap, MB, MBF,
                                           p("ur"), p("dp"))
            return
        ln["exl"].extract_batch_device(d_l.data_ptr(), S, H, W, W, W * H, p("kl"), p("dl"), cap, p("cl"))
        ln["exr"].extract_batch_device(d_r.data_ptr(), S, H, W, W, W * H, p("kr"), p("dr"), cap, p("cr"))
        # all S pairs in one launch; the keypoint counts stay on the device (no host round trip inside a step)
        capi.stereo_match_batch_device(ln["exl"], ln["exr"], 0, 0, S, p("kl"), p("dl"), p("cl"), p("kr"), p("dr"), p("cr"), cap, MB, MBF,
                                       p("ur"), p("dp"))

    def sync_all():
        for ln in lanes:
            ln["exl"].sync(); ln["exr"].sync()

    for i in range(max(args.warmup, 4) * n_lanes):
        step(i)
        if i < 4 * n_lanes:                        # a caller that synchronises lets the strip lengths settle (orb_check_status
            sync_all()                             # shortens a level's FAST strips by one cell per overflowing sync)
    sync_all(); torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all(); torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = shard.max_over_ranks(dist, time.perf_counter() - t0, comm_dev)
    pairs = shard.sum_over_ranks(dist, S * args.steps, comm_dev)
    exl.set_profiling(True)
    for _ in range(4):
        step(0)
    sync_all()
    stage_ms = exl.stage_ms()
    exl.set_profiling(False)
    nl, nr = cl.cpu().numpy(), cr.cpu().numpy()
    if rank != 0:
        return None
    # content statistics of the last synchronised batch, per handle: FAST strips that overflowed their candidate queue (redone by
    # k_fast_strips_dense: same results, more time), strips per frame, candidates per level of frame 0
    content = {}
    for side, e in (("left", exl), ("right", exr)):
        ovf, spf = e.fast_overflows()
        content[side] = {"fast_strips_overflowed_per_level": [int(v) for v in ovf], "fast_strips_per_frame_per_level": [int(v) for v in spf],
                         "fast_candidates_per_level_frame0": [int(v) for v in e.level_counts(0)[1]]}
    pyr_px = sum(int(exl.pyramid_level(0, l).size) for l in range(8))
    u = ur.cpu().numpy().reshape(S, cap)
    z = dp.cpu().numpy().reshape(S, cap)
    mean_kp = float(nl.mean())
    bytes_img = algorithmic_bytes_per_frame(W, H, pyr_px, mean_kp)
    # per pair: both extractions + the stereo search's reads (both keypoint/descriptor sets, 11x(2L+11)-px SAD bands)
    bytes_pair = 2 * bytes_img + int(nl.mean() + nr.mean()) * 60 + int(nl.mean()) * 2 * 11 * 21
    # counters of the dominant kernel measured live, like configs 4 and 5: child runs of this configuration under rocprofv3 --pmc
    c3_child = ["--config", "c3", "--steps", "3", "--warmup", "3", "--no-cpu-baseline", "--no-live-traffic"] + (["--c3-two-handles"] if not one else [])
    rf, rv = roofline_blocks(stage_ms, 2 * S if one else S, bytes_img, float(stage_ms[4]), live=(world == 1 and not args.no_live_traffic),
                             child_args=c3_child)
    rf["algorithmic_bytes_per_pair"] = int(bytes_pair)
    out = {"metric": "frames/sec ORB extract + stereo search, 1241x376 stereo pairs, 8-level 2000-feat; HBM GB/s vs peak",
           "value": round(2 * pairs / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8", "data": "synthetic",
           "config": {"workload": "BASELINE configs[2]: %d synthetic KITTI-sized stereo pairs (1241x376, disparity 12+8*floor(y/94), "
                                  "nFeatures 2000) per step: extract left + right%s, ComputeStereoMatches of all pairs (one launch) on the "
                                  "device pyramids; %d independent lanes of handles and buffers that consecutive steps alternate between"
                                  % (S, " as ONE batch of 2 S frames through one handle" if one else " through two handles on two streams", n_lanes),
                      "lanes": n_lanes, "pairs_per_step": S, "pairs_per_s": round(pairs / elapsed, 2),
                      "ms_per_pair": round(elapsed / (pairs / world) * 1e3, 4), "mean_keypoints_left": round(mean_kp, 1),
                      "mean_stereo_matches": round(float((u[:, :] >= 0).sum() / S), 1),
                      "stage_ms_per_launch_left_handle": {n: round(float(v), 4) for n, v in zip(STAGES + ["extract_total"], stage_ms)},
                      "content_stats": content},
           "roofline": rf}
    if rv:
        out["roofline_valu"] = rv
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle
        all_cores = pin_one_core()
        rl, rr = oracle.Extractor(nf), oracle.Extractor(nf)
        kg_l = kl.cpu().numpy().view(capi.KP_DTYPE).reshape(S, cap)
        dg_l = dl.cpu().numpy().reshape(S, cap, 32)
        ts, parity = [], True
        for i in range(min(S, 2 + max(1, args.cpu_frames // 10))):
            t0 = time.perf_counter()
            k1, d1 = rl.extract(lefts[i])
            k2, d2 = rr.extract(rights[i])
            uo, zo = oracle.stereo_matches(rl, rr, k1, d1, k2, d2, MB, MBF)
            dt = time.perf_counter() - t0
            if i >= 2:
                ts.append(dt)
            n = int(nl[i])
            parity &= n == len(k1) and kg_l[i, :n].tobytes() == k1.tobytes() and np.array_equal(dg_l[i, :n], d1)
            parity &= u[i, :n].tobytes() == uo.tobytes() and z[i, :n].tobytes() == zo.tobytes()
        if all_cores:
            os.sched_setaffinity(0, set(all_cores))
        med = statistics.median(ts)
        out["cpu_baseline"] = {"value": round(2.0 / med, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d of the benchmark pairs after 2 warm-up pairs: oracle extract x2 + stereo search, one "
                                         "pinned thread, median" % len(ts), "ms_per_pair_median": round(med * 1e3, 3),
                               "gpu_matches_oracle_on_sample": bool(parity)}
    return out


# ================================================================== config c5: stream vs keyframe DB
def run_c5(args, rank, local_rank, world, dev, comm_dev, dist):
    """EuRoC-sized stream: a step = ONE 752x480 query frame: extract + vocabulary descent + SearchByBoW against every
    keyframe of a 1000-keyframe descriptor DB resident in HBM (the Relocalization candidate loop, reference
    src/Tracking.cc:1471-1492, is the batch axis); the extractor stream runs ahead of the matcher stream over a ring of
    NSLOT query slots (per-slot events: match(i) waits for extract(i), extract(i) for match(i - NSLOT))."""
    W, H, n_kf = 752, 480, 1000
    if os.environ.get("ORB_BENCH_C5_KF"):                      # experiments only (tools/experiments/c5_host_bound.sh): NOT BASELINE's configs[4]
        n_kf = max(8, int(os.environ["ORB_BENCH_C5_KF"])) // 8 * 8
    ex, mt = capi.Extractor(args.nfeatures, device=local_rank), capi.Matcher(0.7, True, device=local_rank)
    cap = ex.max_keypoints
    QMAX = 8                                                      # stream frames per step in the mini-batch variant
    NSLOT = max(2, min(16, args.c5_slots))                        # query slots of up to QMAX stream frames each
    F = n_kf + NSLOT * QMAX
    buf = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)
    d_kps, d_desc = buf(F * cap * 28, torch.uint8), buf(F * cap * 32, torch.uint8)
    d_counts, d_node = buf(F, torch.int32), buf(F * cap, torch.int16)
    valid_np = np.stack([synth.synth_valid_flags(cap, 7000 + i) for i in range(F)])
    d_valid = torch.from_numpy(valid_np).to(dev)
    tree = trained_vocabulary(ex, W, H)
    voc = capi.Vocabulary(tree, device=local_rank)
    n_nodes = voc.level_nodes(4)
    # the keyframe DB: 1000 frames of a moving-camera sequence (125 scenes x 8 views); the stream revisits those scenes
    for k0 in range(0, n_kf, 40):
        n = min(40, n_kf - k0)
        d_b = torch.from_numpy(synth.synth_sequence(k0, n, W, H)).to(dev)
        ex.extract_batch_device(d_b.data_ptr(), n, H, W, W, W * H, d_kps.data_ptr() + k0 * cap * 28,
                                d_desc.data_ptr() + k0 * cap * 32, cap, d_counts.data_ptr() + k0 * 4)
        ex.sync()
    d_ckeys, d_cstart, d_ccnt = buf(F * cap, torch.int32), buf(F * n_nodes, torch.int16), buf(F * n_nodes, torch.int16)
    d_cdesc = buf(F * cap * 32, torch.uint8)                      # descriptors in feature-vector order (orb_featstore.csr_desc)
    voc.transform_device(mt, d_desc.data_ptr(), d_counts.data_ptr(), n_kf, cap, 4, d_node_of=d_node.data_ptr())
    mt.build_csr_desc_device(d_node.data_ptr(), d_counts.data_ptr(), d_desc.data_ptr(), n_kf, cap, n_nodes, d_ckeys.data_ptr(),
                             d_cstart.data_ptr(), d_ccnt.data_ptr(), d_cdesc.data_ptr())
    mt.sync()
    n_q = max(16, args.stream_frames)
    q_first = 3 + 8 * (rank % 100)
    # the stream: 8 consecutive views of a scene of the DB (other noise), then a hop to another scene
    stream_np = np.concatenate([synth.synth_sequence(8 * ((q_first + 61 * g) % (n_kf // 8)), 8, W, H, noise=5)
                                for g in range((n_q + 7) // 8)])[:n_q]
    stream = torch.from_numpy(stream_np).to(dev)
    Q = [1]                                                       # stream frames per step (1 = the per-frame configuration)
    kf_idx = torch.arange(n_kf, dtype=torch.int32, device=dev)
    f_idx = [torch.arange(QMAX, dtype=torch.int32, device=dev) + (n_kf + s * QMAX) for s in range(NSLOT)]
    kf_pairs = kf_idx.repeat(QMAX)
    f_pairs = [f.repeat_interleave(n_kf) for f in f_idx]
    d_match = [buf(QMAX * n_kf * cap, torch.int32) for _ in range(NSLOT)]
    d_nm = [buf(QMAX * n_kf, torch.int32) for _ in range(NSLOT)]
    store = dict(desc=d_desc.data_ptr(), kps=d_kps.data_ptr(), valid=d_valid.data_ptr(), counts=d_counts.data_ptr(),
                 node_of=d_node.data_ptr(), cap=cap, n_frames=F, n_nodes=n_nodes, csr_keys=d_ckeys.data_ptr(),
                 csr_start=d_cstart.data_ptr(), csr_cnt=d_ccnt.data_ptr(), csr_desc=d_cdesc.data_ptr())
    torch.cuda.synchronize()

    # the extraction of one frame is a chain of six latency-bound launches (~92 us for a handful of workgroups each): frame
    # i + 1 goes through a second extractor handle (its own stream and scratch) while frame i is still in its chain
    pads = [int(v) for v in os.environ.get("ORB_BENCH_STREAM_PADS", "0,0").split(",")]     # experiment: see hip_pad_streams
    hip_pad_streams(pads[0])
    exs = [ex] + [capi.Extractor(args.nfeatures, device=local_rank) for _ in range(max(1, args.c5_extractors) - 1)]
    NEX = len(exs)

    # Frame::ComputeBoW (descent 7 us + feature vector 6 us) runs in front of the frame's search on the searching stream.
    # --c5-bow-early moves it behind the frame's extraction, onto a matcher handle of its own per extractor (ordered behind that
    # extractor's stream), so that the searching stream carries nothing but the 1000 matchings -- measured: 0.0947 against
    # 0.0841 ms per frame, the extra cross-stream wait costs more than the 13 us it takes off the searching stream
    bow_early = args.c5_bow_early
    mbs = [capi.Matcher(0.7, True, device=local_rank) for _ in exs] if bow_early else []

    def compute_bow(m, f0, q):
        voc.transform_device(m, d_desc.data_ptr() + f0 * cap * 32, d_counts.data_ptr() + f0 * 4, q, cap, 4,
                             d_node_of=d_node.data_ptr() + f0 * cap * 2)
        m.build_csr_desc_device(d_node.data_ptr() + f0 * cap * 2, d_counts.data_ptr() + f0 * 4, d_desc.data_ptr() + f0 * cap * 32, q, cap,
                                n_nodes, d_ckeys.data_ptr() + f0 * cap * 4, d_cstart.data_ptr() + f0 * n_nodes * 2,
                                d_ccnt.data_ptr() + f0 * n_nodes * 2, d_cdesc.data_ptr() + f0 * cap * 32)

    def extract(i):                                # stream frames i*Q .. i*Q+Q-1 -> query slot i % NSLOT
        s, q = i % NSLOT, Q[0]
        f0 = n_kf + s * QMAX
        exs[i % NEX].extract_batch_device(stream.data_ptr() + ((i * q) % (n_q - q + 1)) * W * H, q, H, W, W, W * H, d_kps.data_ptr() + f0 * cap * 28,
                                d_desc.data_ptr() + f0 * cap * 32, cap, d_counts.data_ptr() + f0 * 4)
        if bow_early:
            mbs[i % NEX].wait_for(exs[i % NEX].stream)
            compute_bow(mbs[i % NEX], f0, q)

    # likewise the matcher side: descent (7 us) + feature vector (6 us) of frame i + 1 run beside frame i's 1000 matchings
    hip_pad_streams(pads[1])
    mts = [mt] + [capi.Matcher(0.7, True, device=local_rank) for _ in range(max(1, args.c5_matchers) - 1)]
    NMT = len(mts)

    def match(i, between=None):
        s, q = i % NSLOT, Q[0]
        f0 = n_kf + s * QMAX
        mt = mts[i % NMT]
        if not bow_early and between is None and not args.c5_pair_kernel and not args.c5_three_calls:
            # Frame::ComputeBoW + the 1000 matchings in ONE call of the C ABI (orb_bow_query_frames_device): what a C++
            # caller's loop body costs the host, not three trips through the interpreter
            mt.bow_query_frames_device(voc, store, f0, q, 4, kf_idx.data_ptr(), n_kf, f_idx[s].data_ptr(), d_match[s].data_ptr(), d_nm[s].data_ptr())
            return
        if not bow_early:
            compute_bow(mt, f0, q)
        if between is not None:
            between()
        if args.c5_pair_kernel:                    # round 3's form: one workgroup per (keyframe, frame) pair
            mt.match_bow_batch_device(store, kf_pairs.data_ptr(), f_pairs[s].data_ptr(), q * n_kf, d_match[s].data_ptr(), d_nm[s].data_ptr())
        else:                                      # one query against many keyframes (csrc/orb_matcher_query.hip)
            mt.match_bow_query_device(store, kf_idx.data_ptr(), n_kf, f_idx[s].data_ptr(), q, d_match[s].data_ptr(), d_nm[s].data_ptr())

    # (the stream a slot's "extracted" event is recorded on: the ComputeBoW handle's when it runs behind the extraction)
    ex_ss = [torch.cuda.ExternalStream((mbs[k] if bow_early else exs[k]).stream, device=dev) for k in range(NEX)]
    ex_wait = [torch.cuda.ExternalStream(e.stream, device=dev) for e in exs]
    mt_ss = [torch.cuda.ExternalStream(m.stream, device=dev) for m in mts]
    ev_ex, ev_mt = [torch.cuda.Event() for _ in range(NSLOT)], [torch.cuda.Event() for _ in range(NSLOT)]

    def run(n, i0=0):
        # (called with both streams idle.)  The two streams are coupled per slot only: with two slots and whole-stream waits
        # the chain match(i-1) -> extract(i+1) -> match(i+1) made two steps cost extract + match + two cross-stream waits
        # (0.150 ms per frame); with the ring the step is the busier stream's chain
        extract(i0)
        ev_ex[i0 % NSLOT].record(ex_ss[i0 % NEX])
        for i in range(i0, i0 + n):
            j = i + 1
            if j - NSLOT >= i0:
                ex_wait[j % NEX].wait_event(ev_mt[j % NSLOT])      # match(j - NSLOT) has let go of the slot
            extract(j)
            ev_ex[j % NSLOT].record(ex_ss[j % NEX])
            mt_ss[i % NMT].wait_event(ev_ex[i % NSLOT])
            match(i)
            ev_mt[i % NSLOT].record(mt_ss[i % NMT])

    # The same loop from C (tools/c5_loop.c, built on demand): the reference's caller is C++ (Tracking::Relocalization), and
    # through the interpreter the submission of a step -- two library calls, four event operations -- is what bounds it once
    # the GPU side is pipelined.  Only the plain per-frame configuration (the fused ComputeBoW + search call) has a C form.
    c_loop = None
    plain = not (bow_early or args.c5_pair_kernel or args.c5_three_calls)
    if plain and not args.c5_python_loop:
        c_loop = load_c5_loop()
    if c_loop is not None:
        import ctypes as C

        class LoopArgs(C.Structure):
            _fields_ = [("ex", C.POINTER(C.c_void_p)), ("mt", C.POINTER(C.c_void_p)), ("voc", C.c_void_p), ("store", C.c_void_p),
                        ("d_stream", C.c_void_p), ("d_kf_index", C.c_void_p), ("d_f_index", C.POINTER(C.c_void_p)),
                        ("d_match", C.POINTER(C.c_void_p)), ("d_nmatches", C.POINTER(C.c_void_p)), ("d_kps", C.c_void_p),
                        ("d_desc", C.c_void_p), ("d_counts", C.c_void_p)] + \
                       [(n, C.c_int32) for n in ("n_ex", "n_mt", "n_slots", "slot_stride", "n_kf", "cap", "rows", "cols", "n_stream",
                                                 "levelsup", "check_ori")] + [("ratio", C.c_float)]
        arr = lambda vals: (C.c_void_p * len(vals))(*vals)
        c_store = capi.Matcher._store(store)
        c_keep = (arr([e.h.value for e in exs]), arr([m.h.value for m in mts]), arr([t.data_ptr() for t in f_idx]),
                  arr([t.data_ptr() for t in d_match]), arr([t.data_ptr() for t in d_nm]), c_store)
        la = LoopArgs(c_keep[0], c_keep[1], voc.h, C.cast(C.pointer(c_store), C.c_void_p), stream.data_ptr(), kf_idx.data_ptr(), c_keep[2],
                      c_keep[3], c_keep[4], d_kps.data_ptr(), d_desc.data_ptr(), d_counts.data_ptr(), NEX, NMT, NSLOT, QMAX, n_kf, cap, H, W,
                      n_q, 4, 1, 0.7)

        def run_c(n, i0=0):
            sub = C.c_double(0.0)
            rc = c_loop.c5_loop_run(C.byref(la), i0, n, C.byref(sub))
            if rc != 0:
                raise SystemExit("c5_loop_run failed: %d (%s)" % (rc, capi.lib().orb_last_error().decode(errors="replace")))
            return sub.value
    sync_all = lambda: ([e.sync() for e in exs], [m.sync() for m in mbs + mts], torch.cuda.synchronize())

    for _ in range(3):                             # (synchronised calls first: the FAST strip lengths settle)
        run(1)
        sync_all()

    def timed(loop):
        loop(max(args.warmup, 2))
        sync_all()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        sub = loop(args.steps)
        t_sub = time.perf_counter() - t0           # host side alone: when it is close to `elapsed` the step is bound by the launches
        sync_all()
        if dist is not None:
            dist.barrier()
        return shard.max_over_ranks(dist, time.perf_counter() - t0, comm_dev), (sub if sub is not None else t_sub)

    py_elapsed, py_submit = timed(run)
    elapsed, t_submit = timed(run_c) if c_loop is not None else (py_elapsed, py_submit)
    if c_loop is not None and os.environ.get("C5_LOOP_TIMING"):
        c_loop.c5_loop_report()                    # (stderr: the host's time per call site of the C loop)
    done = shard.sum_over_ranks(dist, args.steps, comm_dev)
    # the same stream in mini-batches of QMAX frames per step (offline sequence processing): throughput, not `value`
    mini_fps = 0.0
    if not args.c5_no_minibatch:
        Q[0] = QMAX
        run(3)
        [e.sync() for e in exs]; [m.sync() for m in mbs + mts]; torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_mb = max(10, args.steps // QMAX)
        run(n_mb)
        [e.sync() for e in exs]; [m.sync() for m in mbs + mts]; torch.cuda.synchronize()
        mini_fps = n_mb * QMAX / (time.perf_counter() - t1)
        Q[0] = 1
    # non-overlapped durations: descent + feature vector + search as the host sees them, and the search kernel alone by HIP
    # events on the matcher's own stream (the dominant kernel of this configuration: k_match_bow_query over 1000 keyframes)
    t_m, t_k = [], []
    N_ALONE = 12
    for i in range(N_ALONE):
        exs[i % NEX].wait_for(mts[i % NMT].stream); extract(i); exs[i % NEX].sync(); [m.sync() for m in mbs]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t1 = time.perf_counter(); match(i, between=lambda: e0.record(mt_ss[i % NMT])); e1.record(mt_ss[i % NMT]); mts[i % NMT].sync()
        t_m.append(time.perf_counter() - t1)
        t_k.append(e0.elapsed_time(e1))
    match_ms = statistics.median(t_m) * 1e3
    kernel_ms = statistics.median(t_k[4:])
    if rank != 0:
        return None
    last, last_q = (N_ALONE - 1) % NSLOT, (N_ALONE - 1) % n_q      # slot / stream frame of the last match above
    fq = n_kf + last * QMAX                        # store index of that stream frame
    nm = d_nm[last][:n_kf].cpu().numpy()
    cnts = d_counts.cpu().numpy()
    n1 = float(cnts[:n_kf].mean())
    bytes_query = n_kf * (n1 * (32 + 4 + 1 + 4) + 8 * n_nodes + 4 * float(cnts[fq]))     # B_bow per pair
    ach = bytes_query / (kernel_ms * 1e-3) / 1e9
    kern = "k_match_bow_store" if args.c5_pair_kernel else "k_match_bow_query"
    traffic, traffic_src, valu = None, None, None
    if world == 1 and not args.no_live_traffic:
        lc = live_counters(kern, 1, child_args=["--config", "c5", "--steps", "24", "--warmup", "4", "--no-cpu-baseline", "--no-live-traffic",
                                                "--c5-no-minibatch"] + (["--c5-pair-kernel"] if args.c5_pair_kernel else []))
        traffic, valu = lc["traffic"], lc["valu"]
        if traffic is not None:
            traffic_src = "measured in this run: child runs of bench.py --config c5 (24 steps) under rocprofv3 --pmc FETCH_SIZE / --pmc " \
                          "WRITE_SIZE, per launch of the search kernel; FETCH x2 (profiles/r02_fetch_calibration.json)"
    if traffic is None:
        tr = load_profile("c5_pmc_traffic.json")
        if tr and kern in tr.get("bytes_per_launch", {}):
            traffic, traffic_src = int(tr["bytes_per_launch"][kern]), "profiles/c5_pmc_traffic.json"
    out = {"metric": "frames/sec ORB extract + SearchByBoW vs 1000-keyframe DB, 752x480 8-level 1000-feat; HBM GB/s vs peak",
           "value": round(done / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8", "data": "synthetic",
           "config": {"workload": "BASELINE configs[4]: 752x480 stream, per frame extract + vocabulary descent + SearchByBoW against "
                                  "a %d-keyframe DB in HBM (moving-camera sequence of 125 scenes x 8 views; the stream revisits "
                                  "them)" % n_kf, "pair_matchings_per_s": round(n_kf * done / elapsed, 0),
                      "mean_matches_per_pair": round(float(nm.mean()), 2), "max_matches_per_pair": int(nm.max()),
                      "distinct_stream_frames": n_q, "extractor_handles": NEX, "matcher_handles": NMT, "compute_bow": "behind the extraction" if bow_early else "in front of the search",
                      "frames_per_s_in_mini_batches_of_%d" % QMAX: round(mini_fps, 1),
                      "transform_plus_match_ms_alone": round(match_ms, 4),
                      "query_slots": NSLOT,
                      "frame_loop": "C (tools/c5_loop.c through ctypes: what a C++ caller of the C ABI submits)" if c_loop is not None else "Python (bench.py)",
                      "process_cpus": len(os.sched_getaffinity(0)),
                      "host_submit_ms_per_step": round(t_submit / args.steps * 1e3, 4),
                      "python_loop_ms_per_step": round(py_elapsed / args.steps * 1e3, 4),
                      "python_loop_host_submit_ms_per_step": round(py_submit / args.steps * 1e3, 4)},
           "roofline": {"bound": "hbm", "kernel": kern, "achieved": round(ach, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "traffic_source": traffic_src, "algorithmic_bytes_per_query": int(bytes_query),
                        "kernel_ms_per_launch": round(kernel_ms, 4), "counters": valu,
                        "binding": "latency: a workgroup's keyframe pair is a chain of barrier-separated stages (staging, phase 1 at the "
                                   "issue rate of 21 instructions per 64 distances, the per-node replay, histogram, write-out; "
                                   "tools/qk_stamps.py), one 16-wave workgroup per CU -- not HBM"}}
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle
        all_cores = pin_one_core()
        ref = oracle.Extractor(args.nfeatures)
        # the DB side of the sample: GPU-extracted keyframes (their parity is the extractor tests' business), 60 keyframes
        sample_kf = list(range(0, n_kf, max(1, n_kf // 60)))[:60]
        kps_all = d_kps.cpu().numpy().view(capi.KP_DTYPE).reshape(F, cap)
        desc_all = d_desc.cpu().numpy().reshape(F, cap, 32)
        mg = d_match[last][:n_kf * cap].cpu().numpy().reshape(n_kf, cap)
        qi = last_q
        t0 = time.perf_counter()
        kq, dq = ref.extract(stream_np[qi])
        t_ext = time.perf_counter() - t0
        parity = len(kq) == int(cnts[fq]) and np.array_equal(desc_all[fq, :len(kq)], dq)
        t0 = time.perf_counter()
        fvq = oracle.featvec_from_nodes(oracle.vocab_transform(tree, dq, 4)[1])
        t_tr = time.perf_counter() - t0
        t_match = 0.0
        for kf in sample_kf:
            n = int(cnts[kf])
            dk, kk = desc_all[kf, :n], kps_all[kf, :n]
            fvk = oracle.featvec_from_nodes(oracle.vocab_transform(tree, dk, 4)[1])     # DB side is precomputed in the reference too
            t0 = time.perf_counter()
            nmr, mr = oracle.search_by_bow(dk, kk["angle"], valid_np[kf][:n], fvk, dq, kq["angle"], fvq, 0.7, True)
            t_match += time.perf_counter() - t0
            parity &= nmr == int(nm[kf]) and np.array_equal(mr, mg[kf, :len(kq)])
        if all_cores:
            os.sched_setaffinity(0, set(all_cores))
        per_frame = t_ext + t_tr + t_match / len(sample_kf) * n_kf
        out["cpu_baseline"] = {"value": round(1.0 / per_frame, 4), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "one stream frame: oracle extract + vocabulary descent timed whole, SearchByBoW timed on %d of "
                                         "the %d keyframes and scaled to %d; one pinned thread" % (len(sample_kf), n_kf, n_kf),
                               "extract_ms": round(t_ext * 1e3, 3), "transform_ms": round(t_tr * 1e3, 3),
                               "match_ms_per_pair": round(t_match / len(sample_kf) * 1e3, 4),
                               "gpu_matches_oracle_on_sample": bool(parity)}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: start the N ranks as fresh child processes BEFORE anything here
        # touches the GPU (no exec from a process that initialised HIP), and leave with their exit code
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        sys.exit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to print a mislabelled number" % (args.gpus, world))
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
    out = {"c4": run_c4, "c3": run_c3, "c5": run_c5}[args.config](args, rank, local_rank, world, dev, comm_dev, dist)
    if rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
