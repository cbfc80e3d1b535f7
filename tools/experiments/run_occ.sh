# Run ON THE GPU BOX: waves per SIMD of k_fast_strips_p (ORB_FAST_OCC=0 unrestricted: 22 LDS-limited waves per CU, spread unevenly;
# 4 / 5 / 6: capped through a claimed register count) on both kinds of content.  DESIGN.md section 9, "occupancy".
mkdir -p gpurun_out/occ
for CONTENT in shapes natural; do for occ in 0 5 4; do
  ORB_FAST_OCC=$occ python bench.py --steps 100 --warmup 10 --content $CONTENT --no-cpu-baseline --no-live-traffic --no-natural --no-host-path > gpurun_out/occ/o${occ}_$CONTENT.json 2> gpurun_out/occ/o${occ}_$CONTENT.err
  echo "occ $occ $CONTENT: $(python tools/show_bench.py gpurun_out/occ/o${occ}_$CONTENT.json | head -2 | tr '\n' ' ' | cut -c1-230)"
done; done
