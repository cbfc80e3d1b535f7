#!/bin/bash
# Run ON THE GPU BOX: does LEAVING ROOM beside the latency-bound pyramid kernel (fewer of its workgroups per CU) let the other
# lanes' issue-bound kernels fill its idle issue cycles?  bench.py (512 x 640x480, 4 lanes) per variant.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
Q="--no-cpu-baseline --no-live-traffic --no-natural --no-host-path --steps 100 --warmup 10"
run() { tag=$1; shift; env "$@" python3 bench.py $Q > gpurun_out/cs_$tag.json 2> gpurun_out/cs_$tag.err && python3 tools/show_bench.py gpurun_out/cs_$tag.json "$tag" || tail -3 gpurun_out/cs_$tag.err; }
run base      ORB_X=0
run pyr32     ORB_PYR_LDSMIN=32
run pyr40     ORB_PYR_LDSMIN=40
run pyr53     ORB_PYR_LDSMIN=53
run pyr64     ORB_PYR_LDSMIN=64
run pyr40f4   ORB_PYR_LDSMIN=40 ORB_FAST_OCC=4
Q="$Q --pipeline 6"; run pyr40l6   ORB_PYR_LDSMIN=40
