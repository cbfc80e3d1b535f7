#!/bin/bash
# Run ON THE GPU BOX: the configurations that rely on several streams overlapping, with the library choosing the hardware queue of
# every handle's stream (csrc/orb_streams.hip) and without (ORB_STREAM_BALANCE=0: HIP's own placement).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
Q="--no-cpu-baseline --no-live-traffic --no-natural --no-host-path --steps 300 --warmup 40"
for bal in 1 0; do
  export ORB_STREAM_BALANCE=$bal
  python3 bench.py $Q > gpurun_out/se_c4.json 2>gpurun_out/se_c4.err; python3 tools/show_bench.py gpurun_out/se_c4.json "balance=$bal b512" | head -1
  python3 bench.py $Q --pipeline 2 > gpurun_out/se_c4.json 2>/dev/null; python3 tools/show_bench.py gpurun_out/se_c4.json "balance=$bal b512 lanes2" | head -1
  python3 bench.py $Q --pipeline 3 > gpurun_out/se_c4.json 2>/dev/null; python3 tools/show_bench.py gpurun_out/se_c4.json "balance=$bal b512 lanes3" | head -1
  python3 bench.py $Q --frames-per-gpu 64 > gpurun_out/se_b64.json 2>/dev/null; python3 tools/show_bench.py gpurun_out/se_b64.json "balance=$bal b64 " | head -1
  for v in "2 6 2" "2 4 2" "3 8 3" "2 6 1"; do set -- $v
    python3 bench.py --config c5 --no-cpu-baseline --no-live-traffic --c5-no-minibatch --steps 400 --warmup 40 --c5-matchers $1 --c5-slots $2 --c5-extractors $3 > gpurun_out/se_c5.json 2>/dev/null
    python3 - "$bal" "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/se_c5.json").read().strip().splitlines()[-1])
print("balance=%s c5 (matchers slots extractors)=%s  ms/frame %.4f  python loop %.4f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["config"]["python_loop_ms_per_step"]))
PY
  done
  python3 bench.py --config c3 --no-cpu-baseline --no-live-traffic --steps 100 --warmup 10 > gpurun_out/se_c3.json 2>/dev/null; python3 - gpurun_out/se_c3.json $bal <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("balance=%s c3 value %.0f" % (sys.argv[2], d["value"]))
PY
done
ORB_STREAM_DEBUG=1 ORB_STREAM_BALANCE=1 python3 bench.py --config c5 --no-cpu-baseline --no-live-traffic --c5-no-minibatch --steps 20 --warmup 4 2>&1 >/dev/null | grep "\[orb\]" | head -12
