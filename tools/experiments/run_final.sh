# Final artefacts of the round (run ON THE GPU BOX): full GPU suite, the latency profile, the bench lines.
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > gpurun_out/final/gpu_tests.log 2>&1; tail -3 gpurun_out/final/gpu_tests.log
bash tools/prof_latency.sh > gpurun_out/final/latency.txt 2>&1; tail -9 gpurun_out/final/latency.txt
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; python tools/show_bench.py gpurun_out/final/bench_default.json 2>/dev/null | head -6
python bench.py --config c5 > gpurun_out/final/bench_c5.json 2> gpurun_out/final/bench_c5.err; python tools/show_bench.py gpurun_out/final/bench_c5.json 2>/dev/null | head -4
python bench.py --config c3 > gpurun_out/final/bench_c3.json 2> gpurun_out/final/bench_c3.err; python tools/show_bench.py gpurun_out/final/bench_c3.json 2>/dev/null | head -4
python bench.py --frames-per-gpu 64 --no-cpu-baseline --no-live-traffic --no-natural --no-host-path > gpurun_out/final/bench_b64.json 2> gpurun_out/final/bench_b64.err; python tools/show_bench.py gpurun_out/final/bench_b64.json 2>/dev/null | head -3
bash tools/prof_stats.sh final_b512 --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic --no-natural --no-host-path --pipeline 1 > gpurun_out/final/b512_kernels.txt 2>&1; cat gpurun_out/final/b512_kernels.txt
