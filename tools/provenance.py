#!/usr/bin/env python3
"""Record which build a set of profiles/ artefacts describes: run HERE (the container with .git), after
copying artefacts of one GPU call into profiles/.
  usage: tools/provenance.py <round tag, e.g. r05> <source hash file written on the GPU box> <artefact> [<artefact> ...]
Adds / updates profiles/<tag>_provenance.json: per artefact the sha256 of the kernel sources that ran (tools/source_hash.py,
computed on the GPU box next to the raw profiler output), the git head and dirty flag of this tree at recording time, and the
date.  tests/test_profiles_fresh.py fails when an artefact of the newest round names another source hash than the tree's."""
import datetime
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from source_hash import source_hash  # noqa: E402


def main():
    tag, hfile, arts = sys.argv[1], sys.argv[2], sys.argv[3:]
    ran = open(hfile).read().strip().split()[0]
    git = lambda *a: subprocess.run(["git", "-C", ROOT] + list(a), capture_output=True, text=True).stdout.strip()
    p = os.path.join(ROOT, "profiles", tag + "_provenance.json")
    rec = json.load(open(p)) if os.path.exists(p) else {"artefacts": {}}
    for a in arts:
        name = os.path.basename(a)
        if not os.path.exists(os.path.join(ROOT, "profiles", name)):
            sys.exit("profiles/%s does not exist" % name)
        rec["artefacts"][name] = {"kernel_source_sha256": ran, "tree_matches": ran == source_hash(), "git_head_at_recording": git("rev-parse", "HEAD"),
                                  "tree_dirty_at_recording": bool(git("status", "--porcelain", "--", "orb-slam2-chinesenotes_amd/csrc", "include")),
                                  "recorded": datetime.datetime.now().strftime("%Y-%m-%d %H:%M")}
    rec["note"] = "kernel_source_sha256 = tools/source_hash.py on the GPU box that produced the artefact"
    json.dump(rec, open(p, "w"), indent=1, sort_keys=True)
    print("updated", p)


if __name__ == "__main__":
    main()
