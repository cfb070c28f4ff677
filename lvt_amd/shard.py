"""Multi-GPU sharding of independent sequences: sequence s -> rank s mod G, no data-path collective.
The only cross-rank traffic is the barrier + MAX-reduce that brackets the timed region of bench.py."""
from __future__ import annotations

import time


def assign_sequences(n_sequences: int, world_size: int, rank: int):
    """sequence ids owned by `rank` (KITTI 00..07 stand-ins <-> GPUs 0..7; G < 8 => ceil(8/G) per GPU)"""
    return [s for s in range(n_sequences) if s % world_size == rank]


def timed_region(fn, dist=None, sync=None, device=None):
    """barrier + device sync on both sides, wall time of fn(), MAX over ranks (the driver's contract)."""
    import torch
    if sync: sync()
    if dist is not None: dist.barrier()
    t0 = time.perf_counter()
    out = fn()
    if sync: sync()
    if dist is not None: dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def aggregate_fps(frames_per_rank: int, world_size: int, elapsed_max: float) -> float:
    """whole-job throughput: frames processed by all ranks / max-over-ranks time"""
    return frames_per_rank * world_size / elapsed_max
