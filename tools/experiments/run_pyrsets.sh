# Run ON THE GPU BOX: the three band-table sets of the pyramid chains (ORB_PYR_SET=batch|few|one) over batch shapes;
# prints the pyramid's share of a step (events, single lane) -- the data behind the variant rule in csrc/orb_extractor.hip.
mkdir -p gpurun_out/pyrsets
for geo in "640 480 1000" "752 480 1000" "1241 376 2000"; do set -- $geo
 for n in 40 64 96 128 192; do
  line="$1x$2 n=$n:"
  for s in batch few one; do
    ORB_PYR_SET=$s python bench.py --frames-per-gpu $n --width $1 --height $2 --nfeatures $3 --steps 60 --warmup 8 --no-match --no-cpu-baseline --no-live-traffic --no-natural --no-host-path > gpurun_out/pyrsets/r.json 2>/dev/null
    line="$line $s $(python -c "import json;d=json.loads(open('gpurun_out/pyrsets/r.json').read().strip().splitlines()[-1]);print('%.4f/%.0fk' % (d['config']['stage_ms_per_launch_single_lane']['pyramid(k_pyr_chain launches)'], d['value']/1e3))")"
  done
  echo "$line"
 done
done
