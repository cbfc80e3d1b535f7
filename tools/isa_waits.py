#!/usr/bin/env python3
"""Audit of compiled kernels for loads that were meant to be in flight together but are waited for one by one (the round-5
finding in k_pyr_chain: control flow around batched loads made the compiler put an s_waitcnt vmcnt(0) behind every load).
Reads the gfx950 assembly that `hipcc -save-temps=obj` leaves (…-hip-amdgcn-amd-amdhsa-gfx950.s) and prints, per kernel, the
global loads, the vmcnt waits by count, and every load that is followed by `s_waitcnt vmcnt(0)` before the next load.
  usage: tools/isa_waits.py file.s [kernel-name-substring]"""
import re
import sys


def main():
    path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    name, rows = None, {}
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name = m.group(1)
            rows[name] = []
            continue
        if name and ln.startswith("\t"):
            rows[name].append(ln.strip())
    for k, ins in rows.items():
        if want not in k or not any(i.startswith("s_endpgm") for i in ins):
            continue
        loads = [i for i, s in enumerate(ins) if s.startswith(("global_load", "flat_load", "buffer_load"))]
        if not loads:
            continue
        lone = 0
        for a, b in zip(loads, loads[1:] + [len(ins)]):
            seg = ins[a + 1:b]
            if any(re.match(r"s_waitcnt.*vmcnt\(0\)", s) for s in seg):
                lone += 1
        waits = {}
        for s in ins:
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", s)
            if m:
                waits[int(m.group(1))] = waits.get(int(m.group(1)), 0) + 1
        print("%-90s insts %5d  loads %3d  loads waited for alone %3d  vmcnt waits %s" % (k[:90], len(ins), len(loads), lone, dict(sorted(waits.items()))))


if __name__ == "__main__":
    main()
