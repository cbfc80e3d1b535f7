# Run ON THE GPU BOX: config 5 with three extractor handles (the chains of three frames in flight) -- with the streams balanced
# over the hardware queues, is the 0.048 ms step the two extraction chains?
cd $GRAFT_REPO_ROOT
Q="--config c5 --no-cpu-baseline --no-live-traffic --c5-no-minibatch --steps 600 --warmup 60"
for v in "--c5-extractors 2 --c5-slots 6" "--c5-extractors 3 --c5-slots 6" "--c5-extractors 3 --c5-slots 9" "--c5-extractors 4 --c5-slots 8" "--c5-extractors 3 --c5-slots 9 --c5-matchers 3"; do
  ORB_STREAM_DEBUG=1 python3 bench.py $Q $v > gpurun_out/c5e.json 2> gpurun_out/c5e.err
  python3 - "$v" <<'PY'
import json, re, sys
d = json.loads(open('gpurun_out/c5e.json').read().strip().splitlines()[-1])
pl = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"role (\d) on hardware queue (-?\d+)", open('gpurun_out/c5e.err').read())]
print('%-52s ms/frame %.4f host %.4f kernel %s placement %s' % (sys.argv[1], d['ms_per_step'], d['config']['host_submit_ms_per_step'], d['roofline'].get('kernel_ms_per_launch'), pl), flush=True)
PY
done
