#!/usr/bin/env python3
"""Per launch shape (kernel, grid) durations of a rocprofv3 kernel trace: tools/trace_shapes.py <run_kernel_trace.csv> [name filter]"""
import collections
import csv
import statistics
import sys

d = collections.defaultdict(list)
flt = sys.argv[2] if len(sys.argv) > 2 else "k_"
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").split("(")[0]
    if flt in n:
        d[(n, r["Grid_Size_X"], r["Grid_Size_Y"], r["LDS_Block_Size"], r["VGPR_Count"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print("%-30s grid %8s x %-4s lds %6s vgpr %4s  n %5d  median %8.1f  min %8.1f  max %8.1f us" % (k + (len(v), statistics.median(v), min(v), max(v))))
