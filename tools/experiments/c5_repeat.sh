cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python3 bench.py --config c5 --no-cpu-baseline --no-live-traffic --c5-no-minibatch --steps 400 --warmup 40 > gpurun_out/c5x.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/c5x.json').read().strip().splitlines()[-1]); print('c5 ms/frame', d['ms_per_step'], 'python', d['config']['python_loop_ms_per_step'], 'kernel', d['roofline']['kernel_ms_per_launch'])"; done
