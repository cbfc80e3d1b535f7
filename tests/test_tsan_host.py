"""The library's host-thread logic under the thread sanitizer, on the CPU ("run host code
under TSan"): csrc/orb_host_threads.h -- the chunk pipeline of orb_extract_batch for large host batches with its copy threads,
the per-device fan-out of orb_multi_* with its error merge, the batch partition -- is HIP-free and reaches the GPU through an ops
object; tools/tsan_host.cpp runs that very code against a fake device (streams = worker threads, events = condition variables)
under -fsanitize=thread.  The reference's threading around this path: two extractor threads per stereo frame (src/Frame.cc:82-85)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "orb-slam2-chinesenotes_amd")


@pytest.fixture(scope="module")
def tsan_binary():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-s", "-C", PKG, "tsan-host"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find -ltsan" in (r.stdout + r.stderr):
        pytest.skip("this g++ has no thread sanitizer runtime")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "tsan_host: pipeline, copy threads, fan-out and partition clean" in r.stdout
    return os.path.join(ROOT, "tools", "tsan_host")


def test_host_thread_logic_is_clean_under_tsan(tsan_binary):
    r = subprocess.run([tsan_binary], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "data race" not in r.stderr


def test_the_harness_notices_a_broken_event_protocol(tsan_binary):
    """With the compute stream NOT waiting for a chunk's upload (TSAN_HOST_BREAK=1) the fake DMA thread and the fake kernel
    thread touch the slot's device buffer unordered: the sanitizer must report it -- i.e. a clean run above means something."""
    r = subprocess.run([tsan_binary], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TSAN_HOST_BREAK="1", TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode != 0
    assert "data race" in r.stderr
