#!/bin/bash
# Run HERE (the container with .git) after tools/prof_round.sh <tag> came back through gpurun: gpurun_out/round_<tag>/ and
# gpurun_out/prof_<tag>_b512/ -> profiles/<tag>_*, + profiles/<tag>_provenance.json (tools/provenance.py).
tag=${1:?tag}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=$ROOT/gpurun_out/round_$tag
P=$ROOT/profiles
set -e
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/gpurun_out/prof_${tag}_b512" 512 "$P/${tag}_b512" > /dev/null
arts="${tag}_b512_kernel_stats.csv ${tag}_b512_pmc_traffic.json ${tag}_b512_valu.json"
for f in bench_default bench_b64 bench_natural bench_c3 bench_c5 bench_c5_full_stream; do
    tail -1 "$R/$f.json" > "$P/${tag}_$f.json"; arts="$arts ${tag}_$f.json"
done
for f in c3_kernel_stats c5_kernel_stats latency pyr_stamps qk_stamps; do
    grep -v "amdgpu.ids" "$R/$f.txt" > "$P/${tag}_$f.txt"; arts="$arts ${tag}_$f.txt"
done
python3 "$ROOT/tools/provenance.py" "$tag" "$R/source_sha256" $arts
ls -la "$P" | grep "${tag}_"
