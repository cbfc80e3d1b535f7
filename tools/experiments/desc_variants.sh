#!/bin/bash
# Run ON THE GPU BOX: bench.py (512 x 640x480, extraction + matching, 4 lanes; quick: no cpu baseline / traffic / natural / host
# path) under the variants of the descriptor stage; prints value and the single-lane stage times of each.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
Q="--no-cpu-baseline --no-live-traffic --no-natural --no-host-path --steps 100 --warmup 10"
run() { tag=$1; shift; env "$@" python3 bench.py $Q > gpurun_out/dv_$tag.json 2> gpurun_out/dv_$tag.err && python3 tools/show_bench.py gpurun_out/dv_$tag.json "$tag" || tail -3 gpurun_out/dv_$tag.err; }
run perkp      ORB_DESC_LEVEL=0
run side79     ORB_DESC_LEVEL=1
run noside79   ORB_DESC_LEVEL_SIDE=0
run side79_1cu ORB_DESC_LEVEL_LDSMIN=82
run side110    ORB_DESC_LEVEL_LDS=110
run side64     ORB_DESC_LEVEL_LDS=64
run side52     ORB_DESC_LEVEL_LDS=52
