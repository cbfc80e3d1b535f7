#!/bin/bash
# Run ON THE GPU BOX: the pyramid chains' existing knobs re-measured after the round-5 staging fix (the balance between staging
# and resampling changed): band height, column tables in LDS, rows per item.  bench.py quick runs, stage times per variant.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
Q="--no-cpu-baseline --no-live-traffic --no-natural --no-host-path --steps 60 --warmup 10"
run() { tag=$1; shift; env "$@" python3 bench.py $Q > gpurun_out/pk_$tag.json 2> gpurun_out/pk_$tag.err && python3 tools/show_bench.py gpurun_out/pk_$tag.json "$tag" | head -2 | tr '\n' ' ' | sed 's/stages/\n   stages/' ; echo; }
run base    ORB_X=0
run band12  ORB_PYR_BAND=12
run band20  ORB_PYR_BAND=20
run band24  ORB_PYR_BAND=24
run band32  ORB_PYR_BAND=32
run xqlds   ORB_PYR_XQLDS=1
run rg2     ORB_PYR_RG=2
run b24xq   ORB_PYR_BAND=24 ORB_PYR_XQLDS=1
