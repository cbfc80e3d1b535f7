"""CPU: (1) the oracle's outputs for the first frames of every BASELINE configuration equal the committed sha256 pins
(tests/golden/oracle_pins.json, made by tests/golden/make_oracle_pins.py) -- a drift guard, NOT a pin against OpenCV;
(2) the oracle and the extractor's host-only set-up code run clean under AddressSanitizer + UBSan."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pins_module():
    spec = importlib.util.spec_from_file_location("make_oracle_pins", os.path.join(ROOT, "tests", "golden", "make_oracle_pins.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_oracle_outputs_equal_committed_pins():
    mod = _pins_module()
    want = json.load(open(mod.PINS))
    got = mod.compute()
    assert set(got) == set(want)
    for k in sorted(want):
        assert got[k] == want[k], k


def test_oracle_under_asan_ubsan():
    """oracle/ built with -fsanitize=address,undefined and driven through the same Python wrapper: the pin computation
    (extract at three sizes, both feature-vector builders, SearchByBoW, SearchForInitialization, stereo search) plus the
    oracle's own CPU test files must finish without a sanitizer report and reproduce the pins."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ORBREF_LIB=os.path.join(ROOT, "oracle", "liborbref_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_oracle_pins.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "identical to the committed pins" in r.stdout, (r.stdout[-800:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_matcher.py"),
                        os.path.join(ROOT, "tests", "test_oracle_kat.py")], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def test_extractor_host_setup_code_under_asan_ubsan():
    """csrc/orb_geometry_host.h (tables, strips, path tables, resize tables, slab layout) compiled host-only with
    -fsanitize=address,undefined and swept over 650 sizes / parameter sets with the kernels' invariants checked."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "orb-slam2-chinesenotes_amd"), "asan-geometry"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "no sanitizer report, all invariants hold" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
