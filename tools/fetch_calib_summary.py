#!/usr/bin/env python3
"""profiles/r02_fetch_calibration.json from a rocprofv3 --pmc FETCH_SIZE (csv) run of tools/ubench/fetch_calib:
true bytes / (FETCH_SIZE x 1024) per load width.  usage: fetch_calib_summary.py <rocprof_dir> <out.json>"""
import collections, csv, glob, json, sys

TRUE = {"k_read<unsigned int>": (4, 1 << 30), "k_read<HIP_vector_type<unsigned int, 2u> >": (8, 1 << 30),
        "k_read<HIP_vector_type<unsigned int, 4u> >": (16, 1 << 30), "k_rows48": ("rows48_dword", (1 << 30) // 64 * 48)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"buffer": "1 GiB read once per launch (4x the Infinity Cache)", "true_bytes_over_FETCH_SIZE": {}, "raw": {}}
for name, cs in agg.items():
    if name not in TRUE:
        continue
    w, true = TRUE[name]
    row = {"true_bytes": true}
    for c, v in cs.items():
        row[c] = sum(v) / len(v)
    if "FETCH_SIZE" in row:
        row["FETCH_SIZE_bytes"] = row["FETCH_SIZE"] * 1024
        out["true_bytes_over_FETCH_SIZE"][str(w)] = round(true / row["FETCH_SIZE_bytes"], 4)
    out["raw"][name] = row
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
