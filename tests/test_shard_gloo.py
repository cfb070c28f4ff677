"""CPU tier: the N > 1 path of bench.py (sequence -> rank sharding, barrier + MAX-over-ranks timing) with
world_size 2 on gloo.  No data-path collective exists: ranks only meet at the barrier / timing reduce."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lvt_amd.shard import assign_sequences, aggregate_fps, timed_region


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from parity_util import make_case
    from oracle import pyoracle as O
    seqs = assign_sequences(8, world, rank)
    world_, prm, _ = make_case("kitti", seed=seqs[0], scale=0.25)
    orc = O.Oracle(prm, 1, threads=1)       # stand-in step function on CPU (tests may use the oracle)
    frames = [world_.render_stereo(i) for i in range(3)]

    def body():
        import time
        if rank == 1:
            time.sleep(0.25)                # the slow rank must set the reported time
        return [orc.track(a, b)[1] for a, b in frames]
    dt, poses = timed_region(body, dist=dist)
    ret[rank] = (seqs, dt, float(np.abs(poses[-1]).sum()), aggregate_fps(len(frames), world, dt))
    dist.destroy_process_group()


def test_world_size_2_sharding_and_timing():
    mgr = mp.Manager(); ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    (s0, t0, p0, f0), (s1, t1, p1, f1) = ret[0], ret[1]
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5, 7]           # s mod G, disjoint, covers all 8 sequences
    assert abs(t0 - t1) < 1e-9 and t0 >= 0.25                    # MAX over ranks, identical on every rank
    assert p0 != p1                                              # different sequences were processed
    assert abs(f0 - 2 * 3 / t0) < 1e-9 and f0 == f1


def test_assignment_covers_every_gpu_count():
    for g in (1, 2, 4, 8):
        all_ = sorted(s for r in range(g) for s in assign_sequences(8, g, r))
        assert all_ == list(range(8))
        assert max(len(assign_sequences(8, g, r)) for r in range(g)) == 8 // g
