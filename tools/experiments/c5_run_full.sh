cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 bench.py --config c5 --no-live-traffic --steps 400 --warmup 40 > gpurun_out/c5_full.json 2> gpurun_out/c5_full.err; echo "rc=$?"; tail -2 gpurun_out/c5_full.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/c5_full.json").read().strip().splitlines()[-1])
c = d["config"]
print("value", d["value"], "ms/step", d["ms_per_step"], "loop", c["frame_loop"][:20], "host_submit", c["host_submit_ms_per_step"], "python", c["python_loop_ms_per_step"], c["python_loop_host_submit_ms_per_step"])
print("kernel_ms", d["roofline"]["kernel_ms_per_launch"], "cpu", d.get("cpu_baseline", {}).get("gpu_matches_oracle_on_sample"), d.get("cpu_baseline", {}).get("value"))
PY
bash tools/experiments/c5_variants.sh
