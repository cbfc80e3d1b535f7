set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_vocab.py -x -q -m gpu 2>&1 | tail -3
ORB_VOCAB_FORM=quad timeout -k 10 400 python -m pytest tests/test_gpu_vocab.py tests/test_gpu_matcher.py -x -q -m gpu 2>&1 | tail -3
for form in row quad; do
  echo "== $form"
  ORB_VOCAB_FORM=$form bash tools/prof_stats.sh voc_$form --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic --no-natural --no-host-path --pipeline 1 | grep -i "vocab\|match\|csr"
done
