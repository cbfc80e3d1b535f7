# Run ON THE GPU BOX: config 5 several times with the stream placement printed -- is the step time a function of the placement?
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  ORB_STREAM_DEBUG=1 python3 bench.py --config c5 --no-cpu-baseline --no-live-traffic --c5-no-minibatch --steps 400 --warmup 40 > gpurun_out/c5p.json 2> gpurun_out/c5p.err
  python3 - <<'PY'
import json, re
d = json.loads(open('gpurun_out/c5p.json').read().strip().splitlines()[-1])
pl = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"role (\d) on hardware queue (-?\d+)", open('gpurun_out/c5p.err').read())]
print('c5 ms/frame %.4f  host %.4f  python-loop %.4f  kernel %s  placement(role,queue) %s' % (d['ms_per_step'], d['config']['host_submit_ms_per_step'], d['config']['python_loop_ms_per_step'], d['roofline'].get('kernel_ms_per_launch'), pl), flush=True)
PY
done
