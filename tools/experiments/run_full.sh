mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/full_gpu.log 2>&1; tail -4 gpurun_out/full_gpu.log
timeout -k 10 400 python tools/stress_parity.py --extract 600 > gpurun_out/stress_extract.log 2>&1; tail -3 gpurun_out/stress_extract.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; python tools/show_bench.py gpurun_out/bench_default.json 2>/dev/null | head -20 || tail -c 600 gpurun_out/bench_default.json
python bench.py --config c5 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; tail -c 1500 gpurun_out/bench_c5.json | head -c 1500
