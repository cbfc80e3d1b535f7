# Run ON THE GPU BOX: config 5 started under `taskset -c <one cpu>` (every thread of the process, the HIP runtime's included, is
# born on that cpu) against unpinned starts, alternating: step, host submission, and how often the slow mode shows
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do for pin in yes no; do
  cpu=$(( (RANDOM % 120) + 4 ))
  if [ $pin = yes ]; then pre="taskset -c $cpu"; else pre=""; fi
  $pre python3 bench.py --config c5 --no-cpu-baseline --no-live-traffic --c5-no-minibatch --steps 600 --warmup 60 > gpurun_out/c5pin.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('gpurun_out/c5pin.json').read().strip().splitlines()[-1]); print('%-14s ms/frame %.4f host %.4f python loop %.4f' % ('$pre' or 'unpinned', d['ms_per_step'], d['config']['host_submit_ms_per_step'], d['config']['python_loop_ms_per_step']))"
done; done
