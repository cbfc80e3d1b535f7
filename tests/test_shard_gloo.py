"""N>1 path on CPU: world_size-2 gloo.  Checks the pattern broadcast, the frame partition and that a
sharded run (each rank extracting its shard -- with the CPU oracle standing in for the device, which
tests may do) reproduces the unsharded result exactly."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from orbhip import capi, shard, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pat = torch.zeros(1024, dtype=torch.int8)
    if rank == 0:
        pat.copy_(torch.from_numpy(capi.builtin_pattern()))
    shard.broadcast_pattern(dist, pat, 0)
    start, count = shard.frame_range(total, world, rank)
    ex = oracle.Extractor(300, 1.2, 6, 20, 7)
    counts = []
    for i in range(start, start + count):
        k, d = ex.extract(synth.synth_frame(i, 160, 120))
        counts.append((i, len(k), int(d.astype(np.uint64).sum())))
    tot = shard.sum_over_ranks(dist, count, torch.device("cpu"))
    tmax = shard.max_over_ranks(dist, float(rank + 1), torch.device("cpu"))
    q.put((rank, pat.numpy().copy(), counts, tot, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_range_partitions_exactly():
    for total in (0, 1, 7, 64, 511, 512):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = shard.frame_range(total, world, r)
                seen += list(range(s, s + c))
                assert c in (total // world, total // world + 1)
            assert seen == list(range(total))
    assert shard.weak_range(64, 3) == (192, 64)


def test_world2_gloo_broadcast_and_sharded_equals_unsharded():
    import oracle
    world, total = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    builtin = capi.builtin_pattern()
    ex = oracle.Extractor(300, 1.2, 6, 20, 7)
    want = []
    for i in range(total):
        k, d = ex.extract(synth.synth_frame(i, 160, 120))
        want.append((i, len(k), int(d.astype(np.uint64).sum())))
    got = []
    for rank, pat, counts, tot, tmax in res:
        assert np.array_equal(pat, builtin)            # every rank holds rank 0's table
        assert tot == total and tmax == float(world)   # SUM / MAX reductions used by bench.py
        got += counts
    assert got == want
