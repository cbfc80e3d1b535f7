"""The pin kit for a machine that has OpenCV (tools/pin_against_opencv/): what can be checked WITHOUT OpenCV --
the committed expectation vectors equal a fresh run of their script, the kit's C++ frame generator equals the Python one, and
pin_orb.cpp parses against declaration stubs together with the reference's own include/ORBextractor.h.  None of this pins
anything: parity stays unpinned until pin_orb has run against a real OpenCV."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "tools", "pin_against_opencv")
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))
from orbhip import synth  # noqa: E402


def test_committed_vectors_equal_a_fresh_run(tmp_path):
    out = str(tmp_path / "v.bin")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "make_pin_vectors.py"), out])
    assert open(out, "rb").read() == open(os.path.join(ROOT, "tests", "golden", "pin_vectors.bin"), "rb").read()


def test_cpp_frame_generator_equals_the_python_one(tmp_path):
    src = tmp_path / "gen.cpp"
    src.write_text('#include <cstdio>\n#include <cstdlib>\n#include "pin_frames.h"\n'
                   'int main(int c, char** v) { std::vector<uint8_t> f = pin::synth_frame(atoi(v[1]), atoi(v[2]), atoi(v[3]));\n'
                   '  FILE* o = fopen(v[4], "wb"); fwrite(f.data(), 1, f.size(), o); fclose(o);\n'
                   '  printf("%u\\n", pin::row_hash(f.data(), atoi(v[2]))); return 0; }\n')
    exe = str(tmp_path / "gen")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-I" + KIT, str(src), "-o", exe])
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_pin_vectors as mk
    for idx, w, h in [(0, 640, 480), (1, 752, 480), (7, 97, 61)]:
        out = str(tmp_path / "f.bin")
        first_row = int(subprocess.check_output([exe, str(idx), str(w), str(h), out]).split()[0])
        want = synth.synth_frame(idx, w, h)
        assert np.array_equal(np.fromfile(out, np.uint8).reshape(h, w), want), (idx, w, h)
        assert first_row == int(mk.row_hashes(want)[0])                 # the per-row hash of the vectors file, C++ == Python


@pytest.mark.skipif(not os.path.exists("/root/reference/include/ORBextractor.h"), reason="needs the reference tree (absent on the GPU box)")
def test_pin_program_parses_against_declaration_stubs():
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-I" + os.path.join(KIT, "syntax_check"), "-I" + os.path.join(ROOT, "tests", "support"),
                           "-I/root/reference/include", "-I" + KIT, os.path.join(KIT, "pin_orb.cpp")])
