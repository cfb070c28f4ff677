"""CPU tier: the N > 1 path that bench.py really runs -- its own rank body (`bench.run_rank`: warm-up, the timed region between
barriers, MAX-over-ranks time, SUM of the lost-frame counts, whole-job frames/s, the JSON skeleton) on gloo with world size 2.
Only the per-step work is a stand-in (no GPU here): a backend with the same five methods as bench.HipBackend whose "frames" take a
known time.  No data-path collective exists: the ranks meet at the barrier and at two scalar reductions."""
import os
import socket
import sys
import time

import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lvt_amd.shard import RankEnv, assign_sequences  # noqa: E402


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class StandInBackend:
    """what bench.HipBackend offers to run_rank, with sleeps for steps: rank r's step takes (1 + r) * 5 ms, rank 1 loses one frame"""
    device = None                       # gloo: the reductions run on CPU tensors
    sequences_per_gpu = 1
    workload = "stand-in"

    def __init__(self, env):
        self.env = env
        self.log = []

    def sync(self):
        self.log.append("sync")

    def prepare(self, n_frames):
        self.seqs = assign_sequences(8, self.env.world_size, self.env.rank)
        self.n_frames = n_frames

    def warmup(self, Wm):
        self.log.append(("warmup", Wm))
        return [None] * Wm

    def timed(self, first, K, depth):
        assert first + K == self.n_frames
        for _ in range(K):
            time.sleep(0.005 * (1 + self.env.rank))
        return [(None, None)] * K, (1 if self.env.rank == 1 else 0)

    def extras(self, args, env, warm, poses):
        assert env.rank == 0 and len(poses) == args.steps and len(warm) == args.warmup
        return {"roofline": None, "cpu_baseline": None}


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    from lvt_amd.shard import rank_env
    env = rank_env()
    assert env == RankEnv(rank, rank, world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = bench.parse_args(["--gpus", str(world), "--steps", "12", "--warmup", "3"])
    be = StandInBackend(env)
    res = bench.run_rank(args, env, dist, be)
    ret[rank] = (res, be.seqs, be.log)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_rank_body_on_gloo_world_size_2():
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    (r0, s0, log0), (r1, s1, log1) = ret[0], ret[1]
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5, 7]           # s mod G, disjoint, covers all 8 sequences
    assert r1 is None and r0 is not None                        # rank 0 alone reports
    assert r0["n_gpus"] == 2 and r0["steps"] == 12 and r0["warmup"] == 3 and r0["scaling"] == "weak" and r0["higher_is_better"] is True
    # the slow rank (10 ms per step) sets the time: MAX over ranks, and value = frames of ALL ranks / that time
    assert r0["ms_per_step"] >= 10.0 and r0["ms_per_step"] < 40.0
    assert abs(r0["value"] - 2 * 12 / (r0["ms_per_step"] * 12e-3)) < 0.05 * r0["value"]
    assert r0["tracking"]["frames_not_tracking"] == 1           # SUM over ranks: rank 1's lost frame shows up in rank 0's line
    for log in (log0, log1):                                    # device sync on both sides of the timed region, warm-up first
        assert log[0] == ("warmup", 3) and log.count("sync") == 2
    # rank-local windows, reduced afterwards: the line carries every rank's own rate (rank 0: 5 ms per step, rank 1: 10 ms)
    pr = r0["timing"]["per_rank_fps"]
    assert len(pr) == 2 and 150 < pr[0] <= 200.5 and 75 < pr[1] <= 100.5, pr
    assert abs(r0["value"] - 2 * pr[1]) < 0.01 * r0["value"]     # MAX of the durations = the slow rank's
    for k in ("metric", "value", "unit", "config", "vs_baseline", "dtype", "data", "roofline", "cpu_baseline"):
        assert k in r0


def test_assignment_covers_every_gpu_count():
    for g in (1, 2, 4, 8):
        all_ = sorted(s for r in range(g) for s in assign_sequences(8, g, r))
        assert all_ == list(range(8))
        assert max(len(assign_sequences(8, g, r)) for r in range(g)) == 8 // g


def test_plain_bench_command_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` from a plain shell (no WORLD_SIZE): bench.py launches the two ranks itself under
    torch.distributed.run on 127.0.0.1 and rank 0 prints the one JSON line; rc 0.  The stand-in backend replaces the GPU work,
    everything else -- self-launch, rendezvous, barriers, MAX / SUM reductions -- is the path the driver's 8-GPU run takes."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--backend", "standin"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 12 and r["warmup"] == 3 and r["scaling"] == "weak"
    assert r["ms_per_step"] >= 2.0                               # the slow rank (2 ms per step) sets the time
    assert abs(r["value"] - 2 * 12 / (r["ms_per_step"] * 12e-3)) < 0.05 * r["value"]
    assert r["tracking"]["frames_not_tracking"] == 1             # rank 1's lost frame, summed into rank 0's line


def test_eight_ranks_no_collective_inside_the_window():
    """cfg 5's launch shape on the stand-in backend: 8 ranks, every rank's step takes the same 5 ms, and rank 1 sleeps 50 ms between its
    timed window and the reductions.  With rank-local windows that sleep is nobody's time: the whole-job value equals the sum of the
    ranks' own rates within 1 % (a trailing barrier inside the window would charge the 50 ms -- a quarter of the 200-ms window -- to
    every rank)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "40", "--warmup", "2", "--backend", "standin",
                        "--standin-ms", "5", "--standin-tail-ms", "50"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    t = r["timing"]
    assert r["n_gpus"] == 8 and len(t["per_rank_fps"]) == 8
    # 1 % on an idle box; 5 % here because eight ranks share this container's few cores with whatever else runs (a rank that loses its
    # core for a few ms stretches its own window) -- the charge being excluded is 25 %
    assert abs(r["value"] - t["sum_of_rank_rates"]) <= 0.05 * t["sum_of_rank_rates"], (r["value"], t)
    assert r["ms_per_step"] < 5.9, r["ms_per_step"]                # 5 ms of sleep per step + its overshoot -- not 5 + 50 / 40 = 6.25


def test_cpulist_parser():
    from lvt_amd.shard import parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist("") == [] and parse_cpulist("5") == [5]


def test_total_seqs_spreads_cfg5_over_fewer_gpus():
    import bench
    for g, want in ((1, 8), (2, 4), (4, 2), (8, 1), (3, 3)):
        assert -(-8 // g) == want
    a = bench.parse_args(["--gpus", "2", "--total-seqs", "8"])
    assert a.total_seqs == 8 and a.seqs_per_gpu == 1             # resolved against WORLD_SIZE in main()
