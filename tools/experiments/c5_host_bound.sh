# Run ON THE GPU BOX: is config 5's step the query kernel or the host's submission?  The same loop against keyframe DBs of
# 1000 / 504 / 104 keyframes: the search kernel shrinks with the DB (47 / 24 / 6 us), the 13 API calls per frame do not.
cd $GRAFT_REPO_ROOT
for kf in 1000 504 104 1000; do
  ORB_BENCH_C5_KF=$kf python3 bench.py --config c5 --no-cpu-baseline --no-live-traffic --c5-no-minibatch --steps 400 --warmup 40 > gpurun_out/c5h.json 2> gpurun_out/c5h.err
  python3 - $kf <<'PY'
import json, sys
d = json.loads(open('gpurun_out/c5h.json').read().strip().splitlines()[-1])
print('keyframes %4s: ms/frame %.4f  host submit %.4f  python loop %.4f  search kernel %s' % (sys.argv[1], d['ms_per_step'], d['config']['host_submit_ms_per_step'], d['config']['python_loop_ms_per_step'], d['roofline'].get('kernel_ms_per_launch')), flush=True)
PY
done
