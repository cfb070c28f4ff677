"""Multi-GPU sharding of independent sequences: sequence s -> rank s mod G, one process per GPU, NO data-path
collective ("replicas only": frame k of a sequence depends on the map after frame k-1, sequences share nothing).  The only
cross-rank traffic is the barrier + MAX-reduce that brackets the timed region and a SUM of the lost-frame counts.
bench.py's rank body is built from these helpers; tests/test_shard_gloo.py runs that same rank body on gloo, world size 2."""
from __future__ import annotations

import os
import time
from dataclasses import dataclass


@dataclass
class RankEnv:
    rank: int
    local_rank: int
    world_size: int


def rank_env() -> RankEnv:
    """what torch.distributed.run exports for one-process-per-GPU launches (absent: a single process)"""
    return RankEnv(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def assign_sequences(n_sequences: int, world_size: int, rank: int):
    """sequence ids owned by `rank` (KITTI 00..07 stand-ins <-> GPUs 0..7; G < 8 => ceil(8/G) per GPU)"""
    return [s for s in range(n_sequences) if s % world_size == rank]


def timed_region(fn, dist=None, sync=None, device=None):
    """barrier + device sync on both sides, wall time of fn(), MAX over ranks (the driver's contract).
    The interpreter's cyclic collector is held off while fn() runs: with torch imported a full collection takes ~30 ms -- 300 frame
    periods -- and where it lands depends on the allocation count, not on the frame (found as "frame 433 takes 34 ms" at one
    particular --warmup)."""
    import gc
    import torch
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        if sync: sync()
        if dist is not None: dist.barrier()
        t0 = time.perf_counter()
        out = fn()
        if sync: sync()
        if dist is not None: dist.barrier()
        dt = time.perf_counter() - t0
    finally:
        if gc_was_on: gc.enable()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def sum_over_ranks(value: int, dist=None, device=None) -> int:
    import torch
    if dist is None:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def aggregate_fps(frames_per_rank: int, world_size: int, elapsed_max: float) -> float:
    """whole-job throughput: frames processed by all ranks / max-over-ranks time"""
    return frames_per_rank * world_size / elapsed_max
