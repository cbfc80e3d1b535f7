#!/bin/bash
# Run ON THE GPU BOX (through gpurun, ~8 minutes): every artefact of a round's profiles/ in ONE call, on ONE build -- the
# counter passes of the headline batch, the contract lines of configs 4 / 3 / 5 (+ the 64-frame batch, natural content), the
# kernel tables of configs 3 and 5, the single-frame latency, the pyramid's phase stamps.  Everything lands under
# gpurun_out/round_<tag>/ with the sha256 of the kernel sources that ran (source_sha256); tools/prof_round_collect.sh (run in
# the container afterwards) turns it into profiles/<tag>_* and records the provenance.
#   usage: tools/prof_round.sh <tag, e.g. r05>
tag=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/round_$tag
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python3 tools/source_hash.py > "$OUT/source_sha256"
step() { echo "[prof_round] $*"; }
step "counter passes (b512)"; bash tools/prof_collect.sh ${tag}_b512 > "$OUT/prof_collect.log" 2>&1 || { tail -5 "$OUT/prof_collect.log"; exit 1; }
step "bench default";  python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || { tail -3 "$OUT/bench_default.err"; exit 1; }
step "bench b64";      python3 bench.py --frames-per-gpu 64 --no-cpu-baseline --no-natural --no-host-path --steps 400 --warmup 40 > "$OUT/bench_b64.json" 2> "$OUT/bench_b64.err" || exit 1
step "bench natural";  python3 bench.py --content natural --no-cpu-baseline --no-natural --no-host-path > "$OUT/bench_natural.json" 2> "$OUT/bench_natural.err" || exit 1
step "bench c3";       python3 bench.py --config c3 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err" || { tail -3 "$OUT/bench_c3.err"; exit 1; }
step "bench c5";       python3 bench.py --config c5 --steps 2000 --warmup 200 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err" || { tail -3 "$OUT/bench_c5.err"; exit 1; }
step "bench c5, whole stream"; python3 bench.py --config c5 --steps 3682 --warmup 40 --stream-frames 3682 --no-cpu-baseline --no-live-traffic --c5-no-minibatch > "$OUT/bench_c5_full_stream.json" 2> "$OUT/bench_c5_full_stream.err" || exit 1
step "kernel table c3"; bash tools/prof_stats.sh ${tag}_c3 --config c3 --no-cpu-baseline --no-live-traffic --steps 20 --warmup 6 > "$OUT/c3_kernel_stats.txt" 2>&1 || exit 1
step "kernel table c5"; bash tools/prof_stats.sh ${tag}_c5 --config c5 --no-cpu-baseline --no-live-traffic --c5-no-minibatch --steps 200 --warmup 20 > "$OUT/c5_kernel_stats.txt" 2>&1 || exit 1
step "latency";        bash tools/prof_latency.sh > "$OUT/latency.txt" 2>&1 || { tail -3 "$OUT/latency.txt"; exit 1; }
step "pyramid stamps"; python3 tools/pyr_stamps.py 512 > "$OUT/pyr_stamps.txt" 2>&1 || exit 1
step "query matcher stamps"; python3 tools/qk_stamps.py 1000 30 > "$OUT/qk_stamps.txt" 2>&1 || exit 1
step done
