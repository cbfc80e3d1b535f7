mkdir -p gpurun_out/occ
run() { tag=$1; shift; env "$@" python bench.py --steps 100 --warmup 10 --content $CONTENT --no-cpu-baseline --no-live-traffic --no-natural --no-host-path > gpurun_out/occ/$tag.json 2> gpurun_out/occ/$tag.err; echo "$tag $CONTENT: $(python tools/show_bench.py gpurun_out/occ/$tag.json | head -2 | tr '\n' ' ' | cut -c1-175) $(python -c "import json;d=json.loads(open('gpurun_out/occ/$tag.json').read().strip().splitlines()[-1]);print(d['config'].get('content_stats',{}).get('fast_strips_overflowed_per_level_in_a_settled_batch'))")"; }
for CONTENT in shapes natural; do
run o6c576 ORB_FAST_OCC=6 ORB_FAST_CANDCAP=576
run o6c544 ORB_FAST_OCC=6 ORB_FAST_CANDCAP=544
run o5c640 ORB_FAST_OCC=5
run o5c800 ORB_FAST_OCC=5 ORB_FAST_CANDCAP=800
run o5c960 ORB_FAST_OCC=5 ORB_FAST_CANDCAP=960
done
