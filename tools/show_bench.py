"""Prints the few numbers of a bench.py JSON line that kernel work is steered by: value and the per-stage times."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
tag = sys.argv[2] if len(sys.argv) > 2 else ""
c = d.get("config", {})
print(tag, "value %.0f %s  ms/step %.4f  single-lane %.4f" % (d["value"], d["unit"], d["ms_per_step"], c.get("single_lane_ms_per_step", 0)))
print(tag, " stages", c.get("stage_ms_per_launch_single_lane"), "match", c.get("transform_plus_match_ms_single_lane"))
