#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into profiles/: per-kernel average duration (kernel_stats) and
per-launch HBM-side traffic from separate --pmc FETCH_SIZE / WRITE_SIZE passes.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports
FETCH_SIZE / WRITE_SIZE in KiB, derived from the L2's memory-side request counters (Infinity-Cache hits are
counted).  On gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a 16-B-per-lane streaming read, so it is
DOUBLED for kernels whose loads are 16 B/lane (k_copy_level0: check against its known 307200 B/frame);
kernels that load dwords are left uncorrected and the JSON says so (the guide calls other widths
uncalibrated).  WRITE_SIZE is exact for 16-B stores and taken as is.
usage: pmc_summary.py <stats_dir> <fetch_dir> <write_dir> <frames_per_launch> <out_prefix>"""
import collections
import csv
import glob
import json
import sys

WIDE_LOAD_KERNELS = {"k_copy_level0"}          # 16 B/lane loads -> FETCH_SIZE x2


def kname(n):
    n = n.replace("void ", "")
    return n.split("(")[0].split("<")[0]


def pmc(d, counter):
    rows = list(csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            agg[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    stats_dir, fetch_dir, write_dir, frames, out = sys.argv[1:6]
    frames = int(frames)
    stats = list(csv.DictReader(open(glob.glob(stats_dir + "/*/*kernel_stats.csv")[0])))
    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,avg_us,total_ms,percent\n")
        for r in stats:
            if kname(r["Name"]).startswith("k_"):
                f.write("%s,%s,%.2f,%.3f,%s\n" % (kname(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                   float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    fetch, write = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
    res = {"frames_per_launch": frames, "unit": "bytes per launch (avg over launches)",
           "note": "FETCH_SIZE/WRITE_SIZE in KiB x1024; FETCH doubled only for 16-B/lane kernels (gfx950 correction); "
                   "dword-load kernels uncorrected (uncalibrated width); k_resize_level4 is the average over the 7 levels",
           "bytes_per_launch": {}, "fetch_bytes": {}, "write_bytes": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        fb = fetch.get(k, 0.0) * 1024 * (2 if k in WIDE_LOAD_KERNELS else 1)
        wb = write.get(k, 0.0) * 1024
        res["fetch_bytes"][k] = int(fb)
        res["write_bytes"][k] = int(wb)
        res["bytes_per_launch"][k] = int(fb + wb)
    json.dump(res, open(out + "_pmc_traffic.json", "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
