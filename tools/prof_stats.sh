#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --kernel-trace --stats of one bench.py command; prints the kernel table.
#   usage: tools/prof_stats.sh <tag> [bench.py arguments ...]
tag=${1:?tag}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/stats_$tag
mkdir -p "$OUT"
python3 "$ROOT/tools/source_hash.py" > "$OUT.source_sha256" 2>/dev/null   # which build this output describes (tools/provenance.py)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 "$ROOT/bench.py" "$@" > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").split("(")[0]
        if n.startswith("k_"):
            print("%-28s calls %5s avg_us %9.2f total_ms %9.3f %5s%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
