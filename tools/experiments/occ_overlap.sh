#!/bin/bash
# Run ON THE GPU BOX: with the handles' streams on distinct hardware queues, does leaving room beside FAST (fewer waves per SIMD)
# now let the other lane's latency-bound kernels run beside it?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
Q="--no-cpu-baseline --no-live-traffic --no-natural --no-host-path --steps 200 --warmup 30"
for occ in 5 4 6; do for l in 2 4; do
  ORB_FAST_OCC=$occ python3 bench.py $Q --pipeline $l > gpurun_out/oo.json 2>/dev/null; python3 tools/show_bench.py gpurun_out/oo.json "fast_occ=$occ lanes=$l" | head -1
done; done
for l in 2 4; do ORB_PYR_LDSMIN=40 python3 bench.py $Q --pipeline $l > gpurun_out/oo.json 2>/dev/null; python3 tools/show_bench.py gpurun_out/oo.json "pyr_lds40 lanes=$l" | head -1; done
