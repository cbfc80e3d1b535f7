"""Long-run behaviour.  The reference builds its three extractors once (src/Tracking.cc:117-126) and calls
them once per frame for hours (src/System.cc:221): handles are created and destroyed, the single-frame call is replayed
tens of thousands of times across image sizes (graph re-capture, geometry rebuilds, re-allocation of the pinned staging and
of the scratch slabs when a larger batch comes in between), and neither device memory nor host memory may creep."""
import gc
import os

import numpy as np
import pytest

import oracle
from orbhip import capi, synth

pytestmark = pytest.mark.gpu


def _rss_mb():
    import psutil
    return psutil.Process(os.getpid()).memory_info().rss / 2**20


def _free_mb():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


def test_handles_and_calls_do_not_leak_and_stay_exact():
    import torch
    dev = torch.device("cuda", 0)
    sizes = [(640, 480), (752, 480), (400, 300)]
    frames = {wh: [synth.synth_frame(40 + i, wh[0], wh[1]) for i in range(4)] for wh in sizes}
    ref = oracle.Extractor(1000)
    want = {(wh, i): ref.extract(frames[wh][i]) for wh in sizes for i in range(4)}
    batch_np = synth.synth_batch(900, 64, 640, 480)
    d_batch = torch.from_numpy(batch_np).to(dev)

    def exercise(ex, mt, n_calls, switch_every):
        """n_calls single-frame calls; the size changes every `switch_every` calls"""
        k = 0
        for c in range(n_calls):
            wh = sizes[(c // switch_every) % len(sizes)]
            i = c % 4
            kps, desc = ex.extract(frames[wh][i])
            if c % 997 == 0 or c == n_calls - 1:                     # spot checks along the way, and the last call
                wk, wd = want[(wh, i)]
                assert kps.tobytes() == wk.tobytes() and np.array_equal(desc, wd), (c, wh, i)
                k += 1
        return k

    # warm everything once (allocations that are meant to stay: library, torch context, graph, pinned staging)
    ex, mt = capi.Extractor(1000), capi.Matcher(0.7, True)
    cap = ex.max_keypoints
    d_kps = torch.zeros(64 * cap * 28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(64 * cap * 32, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(64, dtype=torch.int32, device=dev)
    exercise(ex, mt, 30, 3)
    ex.extract_batch_device(d_batch.data_ptr(), 64, 480, 640, 640, 640 * 480, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_cnt.data_ptr())
    ex.sync()
    ex.close(); mt.close()
    gc.collect()
    free0, rss0 = _free_mb(), _rss_mb()

    # 200 create / destroy cycles of both handle kinds, each used once
    for c in range(200):
        e, m = capi.Extractor(1000), capi.Matcher(0.7, True)
        kps, desc = e.extract(frames[sizes[c % 3]][c % 4])
        assert len(kps) == len(want[(sizes[c % 3], c % 4)][0])
        assert m.search_for_initialization(kps, desc, kps, desc, (0.0, 0.0, 64.0 / 640, 48.0 / 480), np.ascontiguousarray(np.stack([kps["x"], kps["y"]], 1), np.float32), 20)[0] > 100
        e.close(); m.close()
    gc.collect()
    free1, rss1 = _free_mb(), _rss_mb()
    assert abs(free1 - free0) < 8, ("device memory after 200 handle cycles", free0, free1)
    assert rss1 - rss0 < 24, ("host RSS after 200 handle cycles", rss0, rss1)

    # 20 000 single-frame calls on one handle: long runs of one size, then a size change every few calls, then every call;
    # a 64-frame device batch every 1 000 calls re-allocates the scratch slabs under the captured graph
    ex, mt = capi.Extractor(1000), capi.Matcher(0.7, True)
    exercise(ex, mt, 60, 3)
    ex.extract_batch_device(d_batch.data_ptr(), 64, 480, 640, 640, 640 * 480, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_cnt.data_ptr())
    ex.sync()
    exercise(ex, mt, 12, 1)
    free2, rss2 = _free_mb(), _rss_mb()
    checks = 0
    done = 0
    for block, (n_calls, switch_every) in enumerate([(1000, 250)] * 17 + [(1000, 10)] * 2 + [(1000, 1)]):
        checks += exercise(ex, mt, n_calls, switch_every)
        done += n_calls
        ex.extract_batch_device(d_batch.data_ptr(), 64, 480, 640, 640, 640 * 480, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_cnt.data_ptr())
        ex.sync()
    assert done == 20000 and checks >= 40
    counts = d_cnt.cpu().numpy()
    k0, dsc0 = ref.extract(batch_np[63])
    got = d_kps.cpu().numpy().view(capi.KP_DTYPE).reshape(64, cap)[63, :counts[63]]
    assert counts[63] == len(k0) and got.tobytes() == k0.tobytes()
    assert np.array_equal(d_desc.cpu().numpy().reshape(64, cap, 32)[63, :counts[63]], dsc0)
    free3, rss3 = _free_mb(), _rss_mb()
    assert abs(free3 - free2) < 8, ("device memory after 20 000 calls", free2, free3)
    assert rss3 - rss2 < 24, ("host RSS after 20 000 calls", rss2, rss3)
    ex.close(); mt.close()
