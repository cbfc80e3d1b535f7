#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the single-frame host call (tools/latency.py), per-kernel average duration and
# the GPU-side span of one call (first kernel start -> last kernel end); then tools/latency_c.sh (the call from C).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lat_trace
mkdir -p "$OUT"
python3 "$ROOT/tools/source_hash.py" > "$OUT.source_sha256" 2>/dev/null   # which build this output describes (tools/provenance.py)
cd /tmp && export TMPDIR=/tmp
ORB_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o run -- python3 "$ROOT/tools/latency.py" 60 > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if r["Kernel_Name"].replace("void ", "").startswith("k_") and not r["Kernel_Name"].replace("void ", "").startswith("k_stream_")]   # (k_stream_*: the queue probes of handle creation, csrc/orb_streams.hip)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# first size only (640x480): calls = groups starting at k_copy_level0
calls, cur = [], []
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
    if n in ("k_copy_level0", "k_pyr_chain") and cur and (n == "k_copy_level0" or cur[-1][0] != "k_pyr_chain"):
        calls.append(cur); cur = []                      # the first kernel of a call: k_copy_level0 or the first k_pyr_chain
    cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
calls.append(cur)
calls = calls[len(calls) // 3: len(calls) // 3 + 40] if len(calls) > 60 else calls[-max(1, len(calls) // 2):]
agg = collections.defaultdict(list)
spans, busy = [], []
for c in calls:
    spans.append((c[-1][2] - c[0][1]) / 1e3)
    busy.append(sum(e - s for _, s, e in c) / 1e3)
    for n, s, e in c:
        agg[n].append((e - s) / 1e3)
print("GPU span per call: %.1f us, sum of kernel durations %.1f us, %d launches" % (sum(spans) / len(spans), sum(busy) / len(busy), len(calls[0])))
for n, v in agg.items():
    print("  %-24s x%d  %.1f us each" % (n, len(v) // len(calls), sum(v) / len(v)))
PY
# the same call from a C caller (no interpreter between the caller and the C ABI)
bash "$ROOT/tools/latency_c.sh" 300
