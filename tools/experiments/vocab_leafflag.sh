# Run ON THE GPU BOX: the descent without the child records of the last level (default) against fetching them (ORB_VOCAB_NO_LEAFFLAG=1)
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_vocab.py tests/test_gpu_matcher_query.py tests/test_gpu_matcher.py -x -q 2>&1 | tail -2
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export ORB_VOCAB_NO_LEAFFLAG=1; else unset ORB_VOCAB_NO_LEAFFLAG; fi
  python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-natural --no-host-path --no-live-traffic > gpurun_out/vr.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('gpurun_out/vr.json').read().strip().splitlines()[-1]); print('no_leafflag=$v: value', d['value'], 'transform+match single lane', d['config']['transform_plus_match_ms_single_lane'])"
done
