set -e
mkdir -p gpurun_out
for cc in 640 512 384 256; do
 for content in shapes natural; do
  ORB_FAST_CANDCAP=$cc python bench.py --steps 60 --warmup 10 --content $content --no-cpu-baseline --no-live-traffic --no-natural --no-host-path > gpurun_out/cc_${cc}_$content.json 2> gpurun_out/cc_${cc}_$content.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/cc_${cc}_$content.json').read().strip().splitlines()[-1])
print("candcap $cc $content", d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['config'].get('content_stats'))
PY
 done
done
