#!/bin/bash
# Run ON THE GPU BOX: the pyramid as TWO launches of longer chains (the few-frames table set) with taller bands, for full batches.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
Q="--no-cpu-baseline --no-live-traffic --no-natural --no-host-path --steps 60 --warmup 10"
run() { tag=$1; shift; env "$@" python3 bench.py $Q > gpurun_out/p2_$tag.json 2> gpurun_out/p2_$tag.err && python3 tools/show_bench.py gpurun_out/p2_$tag.json "$tag" | tr '\n' ' ' | cut -c1-250 || tail -2 gpurun_out/p2_$tag.err; echo; }
run base ORB_X=0
run one_b4  ORB_PYR_SET=one
run one_b8  ORB_PYR_SET=one ORB_PYR_BAND_ONE=8
run one_b12 ORB_PYR_SET=one ORB_PYR_BAND_ONE=12
run one_b16 ORB_PYR_SET=one ORB_PYR_BAND_ONE=16
Q="$Q --pipeline 2"; run lanes2 ORB_X=0
