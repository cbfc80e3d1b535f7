#!/bin/bash
# Run ON THE GPU BOX: config 5 with idle pad streams shifting the hardware queues its four streams land on.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
for a in 0 1 2 3; do for b in 0 1 2 3; do
  ORB_BENCH_STREAM_PADS=$a,$b python3 bench.py --config c5 --no-cpu-baseline --no-live-traffic --c5-no-minibatch --steps 300 --warmup 30 > gpurun_out/c5pad.json 2>/dev/null
  python3 - $a $b <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c5pad.json").read().strip().splitlines()[-1])
print("pads %s,%s  ms/frame %.4f  python loop %.4f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["config"]["python_loop_ms_per_step"]))
PY
done; done
