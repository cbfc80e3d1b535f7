#!/usr/bin/env python3
"""sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h; names and contents, sorted): what a profile artefact must
name to say WHICH build it describes.  tools/prof_*.sh write it next to their raw output on the GPU box (where there is no
.git), tools/pmc_summary.py and tools/provenance.py copy it into profiles/rNN_provenance.json, tests/test_profiles_fresh.py
compares it with the tree.
  usage: tools/source_hash.py            prints the hash of this tree"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_files():
    pk = os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "csrc")
    return sorted(glob.glob(os.path.join(pk, "*.hip")) + glob.glob(os.path.join(pk, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")))


def source_hash():
    h = hashlib.sha256()
    for p in kernel_source_files():
        h.update(os.path.relpath(p, ROOT).encode() + b"\0")
        h.update(open(p, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(source_hash())
