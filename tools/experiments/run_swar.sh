set -e
mkdir -p gpurun_out
for m in 0 1 2; do
  echo "== SWAR $m"
  ORB_FAST_SWAR=$m python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-live-traffic --no-natural --no-host-path > gpurun_out/sw_$m.json 2> gpurun_out/sw_$m.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/sw_$m.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v for k,v in d.items() if 'kernel' in k or 'extract' in k})
print(d.get('config',{}).get('kernel_ms'), d.get('roofline'))
PY
done
for m in 1 2; do
  echo "== parity SWAR $m"
  ORB_FAST_SWAR=$m timeout -k 10 500 python -m pytest tests/test_gpu_extractor.py -x -q -m gpu 2>&1 | tail -3
done
