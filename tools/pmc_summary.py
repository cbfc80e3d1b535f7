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
    agg = collections.defaultdict(list)
    csvs = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    if csvs:
        for r in csv.DictReader(open(csvs[0])):
            if r["Counter_Name"] == counter:
                agg[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    else:                                           # rocprofv3's default rocpd (sqlite) output
        import sqlite3
        db = sqlite3.connect(glob.glob(d + "/*.db")[0])
        for name, val in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            agg[kname(name)].append(float(val))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def kernel_stats(d):
    """[(kernel, calls, avg_ns, total_ns, percent)] from kernel_stats.csv or the rocpd database."""
    csvs = glob.glob(d + "/*/*kernel_stats.csv") + glob.glob(d + "/*kernel_stats.csv")
    if csvs:
        return [(r["Name"], int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]), float(r["Percentage"]))
                for r in csv.DictReader(open(csvs[0]))]
    import sqlite3
    db = sqlite3.connect(glob.glob(d + "/*.db")[0])
    rows = list(db.execute("select name, count(*), avg(duration), sum(duration) from kernels group by name"))
    tot = sum(r[3] for r in rows) or 1.0
    return sorted([(n, c, a, t, 100.0 * t / tot) for n, c, a, t in rows], key=lambda r: -r[3])


def main():
    stats_dir, fetch_dir, write_dir, frames, out = sys.argv[1:6]
    frames = int(frames)
    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,avg_us,total_ms,percent\n")
        for name, calls, avg, tot, pct in kernel_stats(stats_dir):
            if kname(name).startswith("k_"):
                f.write("%s,%d,%.2f,%.3f,%.2f\n" % (kname(name), calls, avg / 1e3, tot / 1e6, pct))
    fetch, write = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
    res = {"frames_per_launch": frames, "unit": "bytes per launch (avg over launches)",
           "note": "FETCH_SIZE/WRITE_SIZE in KiB x1024; FETCH doubled only for 16-B/lane kernels (gfx950 correction); "
                   "dword-load kernels uncorrected (uncalibrated width); resize kernels are averages over their launches (k_resize_pair: levels 1+2, 3+4, 5+6; k_resize_level4p: level 7)",
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
