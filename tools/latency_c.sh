#!/bin/bash
# Run ON THE GPU BOX: the single-frame latency a C caller of the C ABI sees (tools/latency_c.c), beside tools/latency.py
# (the same call through Python/ctypes, which adds ~13 us of interpreter and array handling per call).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-300}
make -s -C "$ROOT/orb-slam2-chinesenotes_amd" latency-c || exit 1
mkdir -p "$ROOT/gpurun_out"
for cfg in "640 480 1000" "752 480 1000" "1241 376 2000"; do
    set -- $cfg
    python3 - "$ROOT" $1 $2 <<'PY'
import sys
sys.path.insert(0, sys.argv[1] + "/orb-slam2-chinesenotes_amd/pyhost")
import numpy as np
from orbhip import synth
w, h = int(sys.argv[2]), int(sys.argv[3])
np.stack([synth.synth_frame(i, w, h) for i in range(8)]).tofile(sys.argv[1] + "/gpurun_out/lat_frames.bin")
PY
    "$ROOT/tools/latency_c" "$ROOT/gpurun_out/lat_frames.bin" $1 $2 8 $3 $N || exit 1
done
rm -f "$ROOT/gpurun_out/lat_frames.bin"
