"""Multi-GPU sharding of independent sequences: sequence s -> rank s mod G, one process per GPU, NO data-path
collective ("replicas only": frame k of a sequence depends on the map after frame k-1, sequences share nothing).  The only
cross-rank traffic is one barrier in front of the timed window and, behind it, the reductions of the rank-local durations and of the lost-frame
counts -- on gloo (CPU tensors), so that RCCL bring-up is not on the failure path of a design that uses no collective.
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


def timed_region(fn, dist=None, sync=None, device=None, tail=None):
    """Rank-local timing of fn(): device sync + barrier BEFORE the window, then every rank times its own call between its own device syncs;
    the durations are reduced AFTERWARDS (MAX = the driver's statistic, MIN, and the list of all of them).  No collective sits inside the
    window: a trailing barrier there would add its own latency and every rank's start-up skew to a window that is 2.4 ms long at the
    driver's --steps 20 -- several per cent of the 6.7 % the >= 7.5 x target leaves.  `tail` (tests) runs between the window and the
    reductions.  Returns (max seconds, fn's result, {"max", "min", "per_rank"}).
    The interpreter's cyclic collector is held off while fn() runs: with torch imported a full collection takes ~30 ms -- 300 frame
    periods -- and where it lands depends on the allocation count, not on the frame (found as "frame 433 takes 34 ms" at one
    particular --warmup)."""
    import gc
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        if sync: sync()
        if dist is not None: dist.barrier()
        t0 = time.perf_counter()
        out = fn()
        if sync: sync()
        dt = time.perf_counter() - t0
    finally:
        if gc_was_on: gc.enable()
    if tail: tail()
    stats = {"max": dt, "min": dt, "per_rank": [dt]}
    if dist is not None:
        import torch
        n = dist.get_world_size()
        t = torch.zeros(n, dtype=torch.float64, device=device)
        t[dist.get_rank()] = dt
        dist.all_reduce(t, op=dist.ReduceOp.SUM)     # (every slot has one writer: the sum IS the gather)
        per = [float(x) for x in t.tolist()]
        stats = {"max": max(per), "min": min(per), "per_rank": per}
    return stats["max"], out, stats


def sum_over_ranks(value: int, dist=None, device=None) -> int:
    import torch
    if dist is None:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gpu_cpu_affinity(local_rank: int):
    """Pin this rank to the CPUs of the NUMA node its GPU hangs off (one busy-polling host thread per rank: it should neither migrate nor
    sit across the socket from its device).  The PCI address comes from torch, the CPU list from sysfs; returns the list that was set, or
    None when the node is unknown (single-socket boxes report -1) or the platform has no sched_setaffinity."""
    try:
        import torch
        p = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(p.pci_device_id))
        base = f"/sys/bus/pci/devices/{bdf}"
        with open(base + "/numa_node") as f:
            if int(f.read().strip()) < 0:
                return None
        with open(base + "/local_cpulist") as f:
            cpus = parse_cpulist(f.read())
        cpus = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return cpus
    except Exception:  # noqa: BLE001 -- affinity is an optimisation, never a reason to fail a run
        return None


def parse_cpulist(text: str):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def aggregate_fps(frames_per_rank: int, world_size: int, elapsed_max: float) -> float:
    """whole-job throughput: frames processed by all ranks / max-over-ranks time"""
    return frames_per_rank * world_size / elapsed_max
