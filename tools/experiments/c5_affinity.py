#!/usr/bin/env python3
"""Run ON THE GPU BOX: does config 5's slow mode (0.060-0.065 instead of 0.049 ms per frame, a whole process long) follow the CPU
the submitting thread runs on?  Prints the GPU's NUMA node / local CPU list, then runs bench.py --config c5 (short form) a few
times unpinned and pinned (taskset) to the GPU-local CPUs and to the others, with the CPU the child started on."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def gpu_local():
    import glob
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            vendor = open(d + "/vendor").read().strip()
            if vendor != "0x1002":
                continue
            out.append((d, open(d + "/numa_node").read().strip(), open(d + "/local_cpulist").read().strip()))
        except OSError:
            pass
    return out

def run(prefix, tag):
    cmd = prefix + [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5", "--no-cpu-baseline", "--no-live-traffic", "--c5-no-minibatch",
                    "--steps", "600", "--warmup", "60"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        print("%-28s ms/frame %.4f  host %.4f" % (tag, d["ms_per_step"], d["config"]["host_submit_ms_per_step"]), flush=True)
    except Exception as e:
        print(tag, "failed", e, p.stderr[-300:], flush=True)

if __name__ == "__main__":
    print("affinity of this process:", sorted(os.sched_getaffinity(0))[:8], "...", len(os.sched_getaffinity(0)), "cpus", flush=True)
    loc = gpu_local()
    for d, node, cpus in loc:
        print(d, "numa", node, "local cpus", cpus)
    try:
        print(subprocess.run(["lscpu"], capture_output=True, text=True).stdout.split("NUMA")[1][:400])
    except Exception:
        pass
    for i in range(4):
        run([], "unpinned %d" % i)
    allowed = sorted(os.sched_getaffinity(0))
    for c in (allowed[0], allowed[len(allowed) // 2], allowed[-1]):
        for i in range(2):
            run(["taskset", "-c", str(c)], "one cpu %d (%d)" % (c, i))
    half = allowed[:len(allowed) // 2], allowed[len(allowed) // 2:]
    for h in half:
        run(["taskset", "-c", ",".join(map(str, h))], "cpus %d-%d" % (h[0], h[-1]))
