mkdir -p gpurun_out/occ
for o in 0 7 6; do
  ORB_DESC_OCC=$o python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-live-traffic --no-natural --no-host-path > gpurun_out/occ/d$o.json 2> gpurun_out/occ/d$o.err
  echo "desc occ $o: $(python tools/show_bench.py gpurun_out/occ/d$o.json | head -2 | tr '\n' ' ' | cut -c1-260)"
done
