import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(f))]
rows.sort()
# take the last 40% of the trace (steady state of the timed loop is in the middle; crude)
n = len(rows)
seg = rows[int(n * 0.3):int(n * 0.6)]
t0, t1 = seg[0][0], seg[-1][1]
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
for s, e, _ in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("kernels", len(seg), "wall_us", (t1 - t0) / 1e3, "gpu_busy_union_us", busy / 1e3, "frac", busy / (t1 - t0))
from collections import Counter
c = Counter(); d = Counter()
for s, e, k in seg: c[k] += 1; d[k] += e - s
for k, v in d.most_common(12): print("%-30s n=%5d avg_us=%8.2f" % (k[:30], c[k], v / c[k] / 1e3))
