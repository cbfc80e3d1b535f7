#!/bin/bash
# Run ON THE GPU BOX: HIP deals streams over GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share one serialise.
# The configurations that rely on several streams overlapping, with 4 / 8 / 16 queues.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
Q="--no-cpu-baseline --no-live-traffic --no-natural --no-host-path --steps 300 --warmup 40"
for q in 4 8 16; do
  export GPU_MAX_HW_QUEUES=$q
  python3 bench.py $Q > gpurun_out/hq_c4_$q.json 2>/dev/null; python3 tools/show_bench.py gpurun_out/hq_c4_$q.json "queues=$q b512" | head -1
  python3 bench.py $Q --frames-per-gpu 64 > gpurun_out/hq_b64_$q.json 2>/dev/null; python3 tools/show_bench.py gpurun_out/hq_b64_$q.json "queues=$q b64 " | head -1
  for v in "2 6 2" "3 8 3" "2 4 2" "3 6 2"; do set -- $v
    python3 bench.py --config c5 --no-cpu-baseline --no-live-traffic --c5-no-minibatch --steps 400 --warmup 40 --c5-matchers $1 --c5-slots $2 --c5-extractors $3 > gpurun_out/hq_c5.json 2>/dev/null
    python3 - "$q" "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/hq_c5.json").read().strip().splitlines()[-1])
print("queues=%s c5 (matchers slots extractors)=%s  ms/frame %.4f  python loop %.4f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["config"]["python_loop_ms_per_step"]))
PY
  done
  python3 bench.py --config c3 --no-cpu-baseline --no-live-traffic --steps 100 --warmup 10 > gpurun_out/hq_c3_$q.json 2>/dev/null; python3 - gpurun_out/hq_c3_$q.json $q <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("queues=%s c3 value %.0f" % (sys.argv[2], d["value"]))
PY
done
