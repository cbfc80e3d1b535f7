"""Batched-frames sharding across the GPUs of one node.

Frames are independent, so the path shards by frame with NO data-path collective.  The only
collective is the broadcast of the 1 KiB BRIEF pattern from rank 0 at start-up (RCCL over xGMI when
the backend is "nccl"; the same code runs on gloo for the CPU tests)."""
import torch


def frame_range(total_frames: int, world: int, rank: int):
    """Contiguous block partition: rank r owns [start, start+count); blocks differ by at most one."""
    base, rem = divmod(total_frames, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def weak_range(frames_per_rank: int, rank: int):
    """Weak scaling used by bench.py: every rank processes its own frames_per_rank synthetic frames."""
    return rank * frames_per_rank, frames_per_rank


def broadcast_pattern(dist, pattern: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast of the int8[1024] pattern tensor (device tensor for nccl, CPU for gloo)."""
    assert pattern.dtype == torch.int8 and pattern.numel() == 1024
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(pattern, src=src)
    return pattern


def max_over_ranks(dist, seconds: float, device) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value: int, device) -> int:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
