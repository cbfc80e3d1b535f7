import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # tests build their device buffers with torch (fills, index tensors: asynchronous on torch's stream) and hand them to the
    # library, whose streams are non-blocking: every *_device call of the ctypes layer first waits for torch's stream
    from orbhip import capi
    capi.ORDER_BEHIND_TORCH = True


@pytest.fixture(scope="session")
def frame0():
    from orbhip import synth
    return synth.synth_frame(0)
