"""profiles/ must describe the build it sits beside: every artefact listed in the newest
profiles/rNN_provenance.json names the sha256 of the kernel sources it was measured on (tools/source_hash.py, computed on the
GPU box); the test fails when that is not the hash of this tree, i.e. when a kernel source changed after the round's profiles
were taken -- regenerate them (tools/prof_collect.sh, tools/provenance.py).  bench.py's fallback counters resolve to the newest
rNN_ artefact by themselves, so no stale copy can stand in for them."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from source_hash import source_hash  # noqa: E402


def _newest_provenance():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_provenance.json")))
    return files[-1] if files else None


def test_newest_round_artefacts_describe_this_tree():
    p = _newest_provenance()
    if p is None:
        pytest.skip("no profiles/rNN_provenance.json yet")
    rec = json.load(open(p))
    cur = source_hash()
    stale = sorted(n for n, a in rec["artefacts"].items() if a["kernel_source_sha256"] != cur)
    assert not stale, "kernel sources changed after these artefacts of %s were measured: %s" % (os.path.basename(p), ", ".join(stale))
    for n in rec["artefacts"]:
        assert os.path.exists(os.path.join(ROOT, "profiles", n)), n


def test_no_unversioned_fallback_copies():
    """profiles/valu.json and profiles/pmc_traffic.json were byte copies of an old round's files that bench.py read as fallback;
    bench.py now takes the newest rNN_b512_* itself."""
    assert not os.path.exists(os.path.join(ROOT, "profiles", "valu.json"))
    assert not os.path.exists(os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    sys.path.insert(0, ROOT)
    import bench
    for name in ("valu.json", "pmc_traffic.json"):
        path = bench.profile_path(name)
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_b512_" + name)))[-1]
        assert path == newest
