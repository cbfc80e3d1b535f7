#!/bin/bash
# Run ON THE GPU BOX: config 5 (752x480 stream, per frame extract + ComputeBoW + SearchByBoW against 1000 keyframes) under
# pipeline variants; prints ms per frame of each.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
Q="--config c5 --no-cpu-baseline --no-live-traffic --c5-no-minibatch --steps 400 --warmup 40"
run() { tag=$1; shift; python3 bench.py $Q "$@" > gpurun_out/c5v_$tag.json 2> gpurun_out/c5v_$tag.err && python3 - gpurun_out/c5v_$tag.json $tag <<'PY' || tail -3 gpurun_out/c5v_$tag.err
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print("%-14s ms/frame %.4f  value %.0f  host_submit %.4f" % (sys.argv[2], d["ms_per_step"], d["value"], c.get("host_submit_ms_per_step", 0)))
PY
}
run m1s4 --c5-matchers 1 --c5-slots 4
run m2s4 --c5-matchers 2 --c5-slots 4
run m2s6 --c5-matchers 2 --c5-slots 6
run m2s8 --c5-matchers 2 --c5-slots 8
run m3s6 --c5-matchers 3 --c5-slots 6
run m3s8 --c5-matchers 3 --c5-slots 8
run m3s8e3 --c5-matchers 3 --c5-slots 8 --c5-extractors 3
run m2s6e3 --c5-matchers 2 --c5-slots 6 --c5-extractors 3
run m4s8e4 --c5-matchers 4 --c5-slots 8 --c5-extractors 4
