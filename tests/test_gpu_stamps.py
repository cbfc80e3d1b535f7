"""The diagnostic phase stamps of the extractor's kernels (orb_extractor_set_pyr_stamps / _qt_stamps / _desc_stamps: thread 0 of a
workgroup leaves the 100 MHz clock at its phase boundaries; tools/pyr_stamps.py, qt_stamps.py, dl_stamps.py print the tables that
DESIGN.md quotes): switching them on and off never changes a result, and what they leave is ordered in time."""
import numpy as np
import pytest

import oracle
from orbhip import capi, synth

pytestmark = pytest.mark.gpu


def _batch(ex, imgs):
    import torch
    B, H, W = imgs.shape
    cap = ex.max_keypoints
    d_img = torch.from_numpy(imgs).cuda()
    d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, W * H, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_cnt.data_ptr())
    ex.sync()
    cnt = d_cnt.cpu().numpy()
    return [(d_kps[f, :cnt[f]].cpu().numpy().tobytes(), d_desc[f, :cnt[f]].cpu().numpy()) for f in range(B)]


def test_stamps_do_not_change_results_and_are_ordered(monkeypatch):
    import torch
    monkeypatch.setenv("ORB_DESC_LEVEL", "1")                       # so that k_desc_level (and its stamps) runs
    monkeypatch.setenv("ORB_DESC_LEVEL_MIN_FRAMES", "1")
    imgs = synth.synth_sequence(300, 40, 640, 480)
    ex = capi.Extractor()
    plain = _batch(ex, imgs)
    ref = oracle.Extractor()
    rk, rd = ref.extract(imgs[7])
    assert plain[7][0] == rk.tobytes() and np.array_equal(plain[7][1], rd)
    B = len(imgs)
    d_p = torch.zeros(B * 64 * 8 * 4, dtype=torch.int64, device="cuda")
    d_q = torch.zeros(B * 8 * 8, dtype=torch.int64, device="cuda")
    d_d = torch.zeros(B * ex.desc_plan()[1] * 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ex.set_pyr_stamps(d_p.data_ptr(), d_p.numel())
    ex.set_qt_stamps(d_q.data_ptr(), d_q.numel())
    ex.set_desc_stamps(d_d.data_ptr(), d_d.numel())
    stamped = _batch(ex, imgs)
    ex.set_pyr_stamps(0, 0); ex.set_qt_stamps(0, 0); ex.set_desc_stamps(0, 0)
    again = _batch(ex, imgs)
    for a, b, c in zip(plain, stamped, again):
        assert a[0] == b[0] == c[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[1], c[1])
    # pyramid: per launch (bands, levels); stamps 0 <= 1 <= 2 <= 3 ... of every workgroup
    p, off = d_p.cpu().numpy(), 0
    layout = ex.pyr_stamp_layout()
    assert len(layout) >= 2
    for bands, steps in layout:
        s = p[off:off + bands * B * 8].reshape(bands * B, 8)
        off += bands * B * 8
        assert (s[:, 0] > 0).all()
        for k in range(2 + steps):
            assert (s[:, k + 1] >= s[:, k]).all()
    # quadtree: workgroup = level * frames + frame
    q = d_q.cpu().numpy().reshape(8, B, 8)
    assert (q[:, :, 0] > 0).all()
    for k in range(4):
        assert (q[:, :, k + 1] >= q[:, :, k]).all()
    assert ((q[:, :, 5] & 0xFFFF) > 0).all()                        # every level of these frames has candidates
    # level-resident descriptor kernel: workgroups with keypoints leave seven ordered stamps
    d = d_d.cpu().numpy().reshape(-1, 8)
    live = d[:, 6] > 0
    assert live.sum() > B
    for k in range(6):
        assert (d[live, k + 1] >= d[live, k]).all()
    ex.close()
