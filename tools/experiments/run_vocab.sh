mkdir -p gpurun_out/voc
timeout -k 10 400 python -m pytest tests/test_gpu_vocab.py tests/test_gpu_matcher.py -x -q -m gpu 2>&1 | tail -3
for v in lds nolds; do
  if [ $v = nolds ]; then export ORB_VOCAB_NO_LDS=1; fi
  bash tools/prof_stats.sh voc_$v --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic --no-natural --no-host-path --pipeline 1 | grep -i "vocab"
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-live-traffic --no-natural --no-host-path > gpurun_out/voc/$v.json 2>/dev/null; echo "$v: $(python tools/show_bench.py gpurun_out/voc/$v.json | head -1)"
done
