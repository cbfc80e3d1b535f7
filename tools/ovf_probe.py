"""Diagnostic (run on the GPU box): per level, how many FAST strips of a 256-frame batch overflowed their candidate queue
and were redone by k_fast_strips_dense, before and after the host shortens that level's strips (orb_get_fast_overflows)."""
import sys; sys.path.insert(0,'orb-slam2-chinesenotes_amd/pyhost')
import torch, numpy as np
from orbhip import capi, synth
ex=capi.Extractor(); cap=ex.max_keypoints; B=256
fr=synth.synth_sequence(0,B); d=torch.from_numpy(fr).cuda()
k=torch.zeros(B*cap*28,dtype=torch.uint8,device='cuda'); de=torch.zeros(B*cap*32,dtype=torch.uint8,device='cuda'); c=torch.zeros(B,dtype=torch.int32,device='cuda')
for it in range(3):
    ex.extract_batch_device(d.data_ptr(),B,480,640,640,640*480,k.data_ptr(),de.data_ptr(),cap,c.data_ptr()); ex.sync()
    print(it, ex.fast_overflows())
