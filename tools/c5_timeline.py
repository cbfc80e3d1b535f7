"""Timeline of a few steady-state steps of `bench.py --config c5` from a rocprofv3 kernel trace: per kernel start (us, relative),
duration and queue, to see whether the per-frame stream is bound by the GPU chain or by the host issuing launches."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].replace("void ", "").startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
seg = rows[n // 3: n // 3 + 60]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("void ", "").split("(")[0][:40]))
