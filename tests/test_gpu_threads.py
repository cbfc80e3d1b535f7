"""The library's host threads from a C++ caller, all at once (tools/threads_stress.cpp): two extractor threads as in the
reference's stereo Frame constructor (src/Frame.cc:82-85), host batches, the multi-device entry and a thread that creates
and destroys handles meanwhile."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_threads_stress_from_a_cpp_caller():
    pkg = os.path.join(ROOT, "orb-slam2-chinesenotes_amd")
    exe = os.path.join(ROOT, "tools", "threads_stress")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tools", "threads_stress.cpp"), "-L" + pkg, "-lorbhip",
                    "-Wl,-rpath," + pkg, "-lpthread"], check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "threads_stress: 0 failures" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
