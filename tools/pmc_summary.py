#!/usr/bin/env python3
"""Summarise a tools/prof_collect.sh output directory (gpurun_out/prof_<tag>/) into tracked files under profiles/:

  <out>_kernel_stats.csv   per-kernel calls / average duration (rocprofv3 --kernel-trace --stats)
  <out>_pmc_traffic.json   per-launch HBM-side bytes from the separate FETCH_SIZE / WRITE_SIZE passes
  <out>_valu.json          per-kernel vector-issue evidence: waves and VALU instructions per launch, VALU-busy,
                           active-lane fraction, LDS instructions / bank-conflict cycles, wait fractions, clock

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports FETCH_SIZE /
WRITE_SIZE in KiB, derived from the L2's memory-side request counters (Infinity-Cache hits are counted).  On
gfx950 FETCH_SIZE reads 1/2 of the bytes of a 16-B-per-lane streaming read; tools/ubench/fetch_calib.hip measured the
same factor 2.0 on a known byte count for 4- and 8-byte-per-lane streams and for 48-byte row gathers (128-byte
requests tallied at 64 B, profiles/r02_fetch_calibration.json), so FETCH_SIZE is DOUBLED for every kernel (without
the calibration file only for 16-B/lane kernels; the JSON says which was used).
WRITE_SIZE is exact for 16-B stores and taken as is.
SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is the sum
over the 8 XCDs.  VALU-busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * kernel cycles) (rocprofv3's VALUBusy).

usage: pmc_summary.py <prof_dir> <frames_per_launch> <out_prefix>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD, N_XCD = 1024, 8
# load width (bytes per lane) of each kernel's dominant global reads -> which calibration factor applies
LOAD_WIDTH = {"k_copy_level0": 16, "k_fast_strips": 16, "k_fast_strips_p": 16, "k_pyr_chain": 16, "k_pyr_chain_p": 16, "k_resize_pair": 8, "k_resize_level4p": 8, "k_orient_desc": "rows48_dword",
              "k_quadtree": 8, "k_match_bow": 16, "k_match_bow_store": 16, "k_bow_assign": 16, "k_vocab_transform": 16, "k_fill_sides": 4}


def kname(n):
    n = n.replace("void ", "")
    return n.split("(")[0].split("<")[0]


def first_batch_dispatch(d):
    """bench.py first extracts a handful of frames to train its vocabulary; the benchmark's own launches start with the
    first k_copy_level0 / k_pyr_chain (the first kernel of a batch) of the full batch size.  Dispatches before that one are left out of every average."""
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if kname(r["Kernel_Name"]) in ("k_copy_level0", "k_pyr_chain")]
    if not rows:
        return 0
    size = lambda r: int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    big = max(size(r) for r in rows)
    return min(int(r["Dispatch_Id"]) for r in rows if size(r) == big)


def counters(d):
    """{kernel: {counter: average per launch}} over the benchmark's own launches"""
    cut = first_batch_dispatch(d)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if int(r["Dispatch_Id"]) >= cut:
                agg[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items() if k.startswith("k_")}


def kernel_stats(d):
    """(kernel, calls, avg ns, total ns, percent) from the kernel trace of the --stats pass, benchmark launches only"""
    cut = first_batch_dispatch(d)
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if int(r["Dispatch_Id"]) >= cut:
                agg[kname(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    tot = sum(sum(v) for k, v in agg.items() if k.startswith("k_")) or 1.0
    return [(k, len(v), sum(v) / len(v), sum(v), 100.0 * sum(v) / tot) for k, v in agg.items()]


def durations(d):
    """average duration (ns) per kernel from a pass's kernel_trace, benchmark launches only"""
    cut = first_batch_dispatch(d)
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if int(r["Dispatch_Id"]) >= cut:
                agg[kname(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    prof, frames, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    calib = {}
    cpath = os.path.join(ROOT, "profiles", "r02_fetch_calibration.json")
    if os.path.exists(cpath):
        calib = json.load(open(cpath))

    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,avg_us,total_ms,percent\n")
        for name, calls, avg, tot, pct in sorted(kernel_stats(prof + "/stats"), key=lambda r: -r[3]):
            if name.startswith("k_"):
                f.write("%s,%d,%.2f,%.3f,%.2f\n" % (name, calls, avg / 1e3, tot / 1e6, pct))

    fetch, write = counters(prof + "/fetch"), counters(prof + "/write")
    res = {"frames_per_launch": frames, "unit": "bytes per launch (avg over launches)",
           "note": "FETCH_SIZE/WRITE_SIZE in KiB x1024; FETCH x2 (gfx950: 128-byte requests tallied at 64 B; calibrated for 4/8/16-byte "
                   "lanes and row gathers in profiles/r02_fetch_calibration.json); resize kernels are averages over their launches",
           "fetch_factor": {}, "bytes_per_launch": {}, "fetch_bytes": {}, "write_bytes": {}}
    for k in sorted(set(fetch) | set(write)):
        w = LOAD_WIDTH.get(k, 4)
        fac = calib.get("hbm_bytes_over_FETCH_SIZE", 2.0 if w == 16 else 1.0)
        fb = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 * fac
        wb = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
        res["fetch_factor"][k] = {"load_bytes_per_lane": w, "factor": fac, "calibrated": bool(calib)}
        res["fetch_bytes"][k] = int(fb)
        res["write_bytes"][k] = int(wb)
        res["bytes_per_launch"][k] = int(fb + wb)
    json.dump(res, open(out + "_pmc_traffic.json", "w"), indent=1)

    sq1, sq2, tcc = counters(prof + "/sq1"), counters(prof + "/sq2"), counters(prof + "/tcc")
    dur1 = durations(prof + "/sq1")
    dur0 = durations(prof + "/stats")
    valu = {"frames_per_launch": frames,
            "note": "per launch averages; quad-cycle counters x4; valu_busy = SQ_ACTIVE_INST_VALU*4/(1024*cycles); "
                    "issue_floor_us = SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * clock): full-rate issue of a wave64 VALU "
                    "instruction takes 2 cycles; clock = GRBM_GUI_ACTIVE/8/duration of the same (profiled) pass",
            "kernels": {}}
    for k, c in sorted(sq1.items()):
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
        d_ns = dur1.get(k, 0.0)
        clock_ghz = cyc / d_ns if d_ns else 0.0
        insts = c.get("SQ_INSTS_VALU", 0.0)
        waves = c.get("SQ_WAVES", 0.0)
        act = c.get("SQ_ACTIVE_INST_VALU", 0.0)
        e = {"duration_us_unprofiled": round(dur0.get(k, 0.0) / 1e3, 2), "duration_us_this_pass": round(d_ns / 1e3, 2),
             "clock_ghz": round(clock_ghz, 3), "waves": int(waves), "valu_insts": int(insts),
             "valu_insts_per_wave": round(insts / waves, 1) if waves else None,
             "valu_busy": round(act * 4 / (N_SIMD * cyc), 4) if cyc else None,
             "active_lane_frac": round(c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (act * 64), 4) if act else None,
             "cycles_per_valu_inst": round(act * 4 / insts, 2) if insts else None,
             "issue_floor_us": round(insts * 2 / N_SIMD / clock_ghz / 1e3, 2) if clock_ghz else None,
             "lds_insts": int(c.get("SQ_INSTS_LDS", 0.0)), "salu_insts": int(c.get("SQ_INSTS_SALU", 0.0)),
             "wave_cycles_per_wave": round(c.get("SQ_WAVE_CYCLES", 0.0) * 4 / waves, 0) if waves else None}
        if e["issue_floor_us"] and d_ns:
            e["issue_floor_frac"] = round(e["issue_floor_us"] / (d_ns / 1e3), 4)
        s2 = sq2.get(k, {})
        if s2:
            wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            e.update({"lds_bank_conflict_cycles": int(s2.get("SQ_LDS_BANK_CONFLICT", 0.0)),
                      "lds_idx_active_cycles": int(s2.get("SQ_LDS_IDX_ACTIVE", 0.0)),
                      "lds_conflict_frac": round(s2.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(s2.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4),
                      "wait_any_frac_of_wave_cycles": round(s2.get("SQ_WAIT_ANY", 0.0) / wc, 4),
                      "wait_inst_any_frac": round(s2.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                      "wait_inst_lds_frac": round(s2.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4),
                      "active_inst_any_frac": round(s2.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
                      "vmem_rd_insts": int(s2.get("SQ_INSTS_VMEM_RD", 0.0))})
        t = tcc.get(k, {})
        if t:
            hit, miss = t.get("TCC_HIT_sum", 0.0), t.get("TCC_MISS_sum", 0.0)
            e.update({"l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None,
                      "ea_rdreq": int(t.get("TCC_EA0_RDREQ_sum", 0.0)), "ea_rdreq_32B": int(t.get("TCC_EA0_RDREQ_32B_sum", 0.0))})
        valu["kernels"][k] = e
    json.dump(valu, open(out + "_valu.json", "w"), indent=1)
    print(json.dumps({k: {x: v[x] for x in ("duration_us_unprofiled", "valu_insts_per_wave", "valu_busy", "active_lane_frac",
                                            "issue_floor_frac", "lds_conflict_frac") if x in v}
                      for k, v in valu["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
