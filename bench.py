#!/usr/bin/env python3
"""bench.py -- stereo frames/sec of the MI355X-native LVT tracking path on KITTI-shaped input.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU) -- and a plain
`python bench.py --gpus N` starts those N ranks itself (same launcher module, 127.0.0.1, a free port) and passes rank 0's line on.
A "step" is one pass of the hot path over one stereo pair of a synthetic KITTI-00-shaped sequence (1241x376,
BASELINE.json configs[1]): `lvt_amd_track_device_async` on a frame resident in HBM, its pose collected with
`lvt_amd_wait_status` (at most --depth frames in flight).  Independent sequences (seed = rank) shard one per GPU, there is
no collective on the data path (lvt_amd/shard.py) -> "scaling": "weak".  Rank 0 prints ONE JSON line.

Beside the headline the line carries (rank 0, N = 1): `se3` = the timed poses against the CPU oracle's on the same frames
(the run FAILS above 1e-4), `sync` = latency of the synchronous entry points the reference's callers use (lvt_track with
host buffers), `batch` = the lock-step batch, `configs` = EuRoC- and TUM-shaped legs, `roofline` = the batched Hamming
matcher against the HBM roof (HIP events on its launch stream, inside this process), `cpu_baseline` = the oracle on this
box's host cores.  The oracle is imported for `se3` / `cpu_baseline` only, outside the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from lvt_amd.shard import aggregate_fps, gpu_cpu_affinity, rank_env, sum_over_ranks, timed_region  # noqa: E402  (no HIP, no torch at import)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s is achievable
POSE_TOL = 1e-4        # BASELINE.json: per-frame SE3 within 1e-4 rel of the CPU reference
METRIC = "stereo frames/sec (KITTI-shaped 1241x376), per-frame SE3 vs CPU ref"


def bmatch(M, N):
    """algorithmic bytes of one masked-Hamming problem: 40(M+N) + N + 16M"""
    return 40 * (M + N) + N + 16 * M


def pose_errors(Rh, th, Ro, to):
    """e_t = |t_gpu - t_cpu| / max(|t_cpu|, 1 m), e_R = angle(R_gpu^T R_cpu)"""
    e_t = float(np.linalg.norm(th - to) / max(np.linalg.norm(to), 1.0))
    e_R = float(np.arccos(np.clip((np.trace(Rh.T @ Ro) - 1) / 2, -1, 1)))
    return e_t, e_R


def live_hamming_traffic(B, M, N, timeout=120, mode=0):
    """HBM counters of the matcher launch, collected NOW: one `rocprofv3 --pmc <counter>` child per counter (FETCH_SIZE and WRITE_SIZE cannot share
    a pass; counters only, no trace domain) over tools/hamming_bench.py with the same B, M, N.  Returns {counter: KB per dispatch, n_counter: dispatches}
    or None (no rocprofv3, this process is itself under a profiler, or the tool failed) -- the caller then falls back to the committed passes."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in (os.environ.get("LD_PRELOAD", "") + os.environ.get("HSA_TOOLS_LIB", "")):
        return None
    out = {}
    tmp = tempfile.gettempdir()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="lvt_pmc_")
        try:
            env = dict(os.environ, TMPDIR=tmp)
            cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "h", "--", sys.executable, os.path.join(HERE, "tools", "hamming_bench.py"),
                   str(B), str(M), str(N), str(mode)]
            subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if ("k_hamming_batched<%d" % mode) in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None
            out[ctr], out["n_" + ctr] = float(np.mean(vals)), len(vals)
        except Exception:  # noqa: BLE001
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out


def pct(a, q):
    return float(np.percentile(np.asarray(a, dtype=np.float64), q))


# =====================================================================================================================
# the rank body: identical for every rank and every backend (tests/test_shard_gloo.py runs it on gloo with a stand-in backend)
# =====================================================================================================================
def run_rank(args, env, dist, backend):
    """W untimed warm-up steps, EXACTLY K timed steps between barrier + device sync on both sides, MAX over ranks; rank 0 gets
    the result dict (None on the others)."""
    K, Wm = args.steps, args.warmup
    backend.prepare(Wm + K)
    warm = backend.warmup(Wm)
    dt, (poses, not_tracking), tstats = timed_region(lambda: backend.timed(Wm, K, args.depth), dist=dist, sync=backend.sync, device=None,
                                                     tail=getattr(backend, "tail", None))
    lost = sum_over_ranks(not_tracking, dist, None)
    if env.rank != 0:
        return None
    S = backend.sequences_per_gpu
    fps = aggregate_fps(K * S, env.world_size, dt)
    result = {
        "metric": METRIC, "value": round(fps, 2), "unit": "frames/s", "n_gpus": env.world_size, "steps": K, "warmup": Wm,
        "ms_per_step": round(1e3 * dt / K, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/f64", "data": "synthetic",
        "config": {"workload": backend.workload + (" [--share-gpu: the ranks share fewer GPUs -- a rehearsal, not a scaling measurement]" if getattr(args, "share_gpu", False) else ""), "sequences_per_gpu": S, "frames_in_flight": args.depth,
                   "total_sequences": S * env.world_size, "total_sequences_requested": (getattr(args, "total_seqs", 0) or None),
                   "parallelism": f"{env.world_size * S} independent sequences, one process per GPU, no collective"},
        "tracking": {"frames_not_tracking": lost, "checked": "lvt_amd_wait_status == 2 on every timed frame of every rank"},
        "timing": {"window": "per rank: device sync, barrier, K steps, device sync -- no collective inside; value = all ranks' frames / MAX of the rank-local durations "
                             "(reduced afterwards on gloo)",
                   "per_rank_fps": [round(K * S / t, 2) for t in tstats["per_rank"]], "sum_of_rank_rates": round(sum(K * S / t for t in tstats["per_rank"]), 2),
                   "ms_per_step_min_rank": round(1e3 * tstats["min"] / K, 4), "cpu_affinity_rank0": (lambda a: None if not a else f"{len(a)} CPUs of the GPU's NUMA node ({a[0]}..{a[-1]})")(getattr(backend, "affinity", None))},
    }
    result.update(backend.extras(args, env, warm, poses))
    return result


# =====================================================================================================================
# the product path on this rank's GPU
# =====================================================================================================================
class HipBackend:
    def __init__(self, args, env):
        import torch
        import lvt_amd
        from lvt_amd.synth import make_world
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the tracking path has no CPU fallback")
        self.torch, self.lvt, self.make_world = torch, lvt_amd, make_world
        dev_index = env.local_rank % torch.cuda.device_count() if args.share_gpu else env.local_rank   # (--share-gpu: a rehearsal of the N-rank path on fewer GPUs)
        torch.cuda.set_device(dev_index)
        self.device = torch.device("cuda", dev_index)
        self.affinity = gpu_cpu_affinity(dev_index) if env.world_size > 1 else None   # (a single rank keeps the scheduler's choice)
        self.env = env
        self.sequences_per_gpu = args.seqs_per_gpu
        self.depth = args.depth
        S = self.sequences_per_gpu
        self.host_frames = S == 1 and not args.device_frames
        self.workload = ("KITTI seq 00-shaped synthetic stereo sequence (1241x376, examples/kitti/vo_config.yaml + calib/00.yml), one sequence per GPU; "
                         + ("frames pre-rendered into PINNED HOST memory (tightly packed, as lvt_track takes them; SURVEY 8d); a step = lvt_amd_track_async "
                            "(the pull of frame t+1 over PCIe overlaps the kernels of frame t) + lvt_amd_wait_status; warm-up runs through the same "
                            "asynchronous entry; K enqueues + K collects inside the timed window, which starts and ends with an idle device"
                            if self.host_frames else
                            "a step = lvt_amd_track_device_async on a frame resident in HBM + lvt_amd_wait_status")) if S == 1 else \
                        (f"{S} independent KITTI seq 00-shaped synthetic stereo sequences per GPU advanced in lock-step (lvt_amd_batch_*; one step = "
                         "one stereo pair of EVERY sequence), frames resident in HBM")

    def sync(self):
        self.torch.cuda.synchronize()

    def _render(self, worlds, n_frames):
        torch = self.torch
        H, W = worlds[0].H, worlds[0].W
        pitch = ((W + 63) // 64) * 64
        fr = torch.zeros((len(worlds), n_frames, 2, H, pitch), dtype=torch.uint8, device=self.device)
        for s, w in enumerate(worlds):
            for i in range(n_frames):
                fr[s, i, :, :, :W] = w.render_stereo_torch(i, device=self.device)
        torch.cuda.synchronize()
        return fr, H, W, pitch

    def prepare(self, n_frames):
        S, rank = self.sequences_per_gpu, self.env.rank
        self.worlds = [self.make_world("kitti", seed=rank * S + s) for s in range(S)]
        self.prm = self.lvt.kitti_params()
        self.n_frames = n_frames
        self.frames, self.H, self.W, self.pitch = self._render(self.worlds, n_frames)
        if S == 1:
            self.vo = self.lvt.LvtSystem.create(self.prm, self.lvt.eSensor_STEREO)
            if self.host_frames:  # the same frames, tightly packed, page-locked: what a caller of lvt_track holds
                torch = self.torch
                self.host = torch.empty((n_frames, 2, self.H, self.W), dtype=torch.uint8).pin_memory()
                self.host.copy_(self.frames[0, :, :, :, :self.W])
                self.sync()
                base, img = self.host.data_ptr(), self.H * self.W
                self.hptr = [(base + (2 * i) * img, base + (2 * i + 1) * img) for i in range(n_frames)]
        else:
            self.vo = self.lvt.LvtBatch(self.prm, S)
            self.lp = [[self.frames[s, i, 0].data_ptr() for s in range(S)] for i in range(n_frames)]
            self.rp = [[self.frames[s, i, 1].data_ptr() for s in range(S)] for i in range(n_frames)]

    def _ptrs(self, i):
        p = self.frames[0, i].data_ptr()
        return p, p + self.H * self.pitch

    def warmup(self, Wm):
        out = []
        if self.sequences_per_gpu == 1 and self.host_frames:  # through the asynchronous host-buffer entry, like the timed frames
            inflight = 0
            for i in range(Wm):
                assert self.vo.track_async_ptr(self.hptr[i][0], self.hptr[i][1], self.H, self.W) == 0
                inflight += 1
                if inflight >= self.depth:
                    out.append(self.vo.wait()); inflight -= 1
            while inflight:
                out.append(self.vo.wait()); inflight -= 1
            return out
        for i in range(Wm):
            if self.sequences_per_gpu == 1:
                l, r = self._ptrs(i)
                out.append(self.vo.track_device(l, r, self.H, self.W, self.pitch))
            else:
                self.vo.track_device_async(self.lp[i], self.rp[i], self.H, self.W, self.pitch)
                out.append(self.vo.wait())
        return out

    def timed(self, first, K, depth):
        """frames are enqueued asynchronously with at most `depth` poses outstanding: the feature stage of frame t+1 overlaps the
        tracking chain of frame t on the device, the host never idles the GPU between frames"""
        vo, H, W, pitch = self.vo, self.H, self.W, self.pitch
        poses, bad, inflight = [], 0, 0
        single = self.sequences_per_gpu == 1

        def collect():
            nonlocal bad
            if single:
                R, t, st = vo.wait_status()
                bad += 0 if st == 2 else 1
            else:
                R, t, st = vo.wait()
                bad += int((st != 2).sum())
            poses.append((R, t))
        host = single and self.host_frames
        for i in range(first, first + K):
            if host:
                vo.track_async_ptr(self.hptr[i][0], self.hptr[i][1], H, W)
            elif single:
                l, r = self._ptrs(i)
                vo.track_device_async(l, r, H, W, pitch)
            else:
                vo.track_device_async(self.lp[i], self.rp[i], H, W, pitch)
            inflight += 1
            if inflight >= depth:
                collect(); inflight -= 1
        while inflight:
            collect(); inflight -= 1
        assert len(poses) == K
        return poses, bad

    # ---- everything beside the headline (rank 0 only, outside the timed region) -----------------------------------------
    def extras(self, args, env, warm, poses):
        out = {"tracking_error_string": self.vo.last_error()}
        if self.sequences_per_gpu != 1:
            out["roofline"], out["cpu_baseline"] = None, None
            return out
        out["config_ordering"] = self.vo.ordering()
        c = self.vo.counts()
        out["tracking_last_frame"] = {"features_left": c["n_left"], "map_size": c["map_size"], "matches": c["n_matches"]}
        out["host_route"] = self.vo.host_stats()
        legs = [("device_resident", self._leg_device_resident), ("kernels", self._leg_kernels), ("roofline", self._leg_roofline)]
        if env.world_size == 1:  # reported at N = 1 only (rank 0's host cores / one GPU to itself)
            legs += [("sync", self._leg_sync), ("batch", self._leg_batch), ("lists_ab", self._leg_lists_ab), ("configs", self._leg_configs)]
            if not args.no_cpu:
                legs += [("cpu", lambda a: self._leg_cpu(a, warm, poses))]
        for name, fn in legs:
            if name in args.skip:
                continue
            try:
                out.update(fn(args))
            except Exception as e:  # noqa: BLE001 -- a broken side leg must not hide the headline; it is reported, not swallowed
                out[name + "_error"] = f"{type(e).__name__}: {e}"
        out.setdefault("roofline", None)
        out.setdefault("cpu_baseline", None)
        return out

    def _leg_device_resident(self, args):
        """the same frames already in HBM (lvt_amd_track_device_async): what the PCIe pull costs the pipeline.  Fresh handle, the same number of
        warm-up and timed frames, the same depth; a side field, never `value`."""
        if not self.host_frames:
            return {}
        n = min(self.n_frames, args.warmup + args.steps)
        Wm = min(args.warmup, n - 1)
        os.environ["LVT_AMD_ORDERING"] = "polling"   # (a handle created beside a live one would order with events: not the path being compared)
        try:
            vo = self.lvt.LvtSystem.create(self.prm, 1)
        finally:
            os.environ.pop("LVT_AMD_ORDERING", None)
        res = {}
        for tag in ("first", "second"):   # (twice on the same handle after a reset: the second pass runs with warm clocks)
            vo.reset()
            inflight, bad = 0, 0
            for i in range(Wm):
                l, r = self._ptrs(i)
                vo.track_device_async(l, r, self.H, self.W, self.pitch); inflight += 1
                if inflight >= self.depth:
                    vo.wait(); inflight -= 1
            while inflight:
                vo.wait(); inflight -= 1
            self.sync()
            t0 = time.perf_counter()
            for i in range(Wm, n):
                l, r = self._ptrs(i)
                vo.track_device_async(l, r, self.H, self.W, self.pitch); inflight += 1
                if inflight >= self.depth:
                    bad += 0 if vo.wait_status()[2] == 2 else 1; inflight -= 1
            while inflight:
                bad += 0 if vo.wait_status()[2] == 2 else 1; inflight -= 1
            self.sync()
            dt = time.perf_counter() - t0
            res[tag] = {"fps": round((n - Wm) / dt, 1), "ms_per_step": round(1e3 * dt / (n - Wm), 4), "frames_not_tracking": bad}
        vo.close()
        return {"device_resident": {"entry": "lvt_amd_track_device_async (frames already in HBM)", "frames": n - Wm, **res}}

    def _leg_kernels(self, args):
        """per-kernel HIP-event times (synchronous mode) over a few frames past the timed ones; the frame's algorithmic bytes"""
        P = min(args.profile_steps, 60)
        if P <= 0:
            return {}
        vo = self.vo
        world = self.worlds[0]
        extra = self.torch.zeros((P, 2, self.H, self.pitch), dtype=self.torch.uint8, device=self.device)
        for k in range(P):
            extra[k, :, :, :self.W] = world.render_stereo_torch(self.n_frames + k, device=self.device)
        self.sync()
        vo.profile_enable(True)
        nl = nr = mp = nm = 0
        for k in range(P):
            p = extra[k].data_ptr()
            vo.track_device(p, p + self.H * self.pitch, self.H, self.W, self.pitch)
            c = vo.counts()
            nl += c["n_left"]; nr += c["n_right"]; mp += c["map_size_at_match"]; nm += c["n_matches"]
        prof = vo.profile_read()
        vo.profile_enable(False)
        nl /= P; nr /= P; mp /= P; nm /= P
        alg = {"k_score": 2.0 * self.W * self.H, "k_brief": (nl + nr) * 40.0, "k_pnp": nm * 32.0 + 56.0}
        work = [x for x in prof if "k_gate" not in x[0] and "wait for" not in x[0]]
        tot = sum(ms for _, ms, _ in work)
        kernels = []
        for name, ms, calls in prof:
            if calls:
                gate = "k_gate" in name or "wait for" in name
                kernels.append({"kernel": name, "avg_us": round(1e3 * ms / calls, 3), "share": None if gate else round(ms / tot, 4)})
        chain = [x for x in work if x[0].startswith(("k_match_map", "k_track_mid", "k_pnp", "k_candidates(staged)", "k_triangulate"))]
        dom = max(chain or work, key=lambda x: x[1])
        dom_us = 1e3 * dom[1] / max(dom[2], 1)
        ab = next((v for k, v in alg.items() if dom[0].startswith(k)), 0.0)
        ach = ab / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
        frame_bytes = 2.0 * self.W * self.H + (nl + nr) * 40.0 + bmatch(mp, nl) + bmatch(nl, nr) + 24.0 * mp + 56.0
        return {"kernels": kernels, "frame_algorithmic_bytes": round(frame_bytes),
                "roofline_pipeline_dominant": {"kernel": dom[0], "bound": "fp64 VALU issue of one CU (Levenberg-Marquardt sweeps), not HBM",
                                               "achieved": round(ach, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 8),
                                               "traffic": None, "avg_us": round(dom_us, 3), "algorithmic_bytes_per_launch": round(ab, 1)}}

    def _leg_roofline(self, args):
        """the batched Hamming matcher at B = 8192 KITTI-nominal problems per launch (962 MB: past the 256-MB Infinity Cache)"""
        torch, lvt = self.torch, self.lvt
        dev, H, W = self.device, self.H, self.W
        B, M, N = args.hamming_batch, 1000, 1500
        g = torch.Generator(device=dev); g.manual_seed(1234)
        qd = torch.randint(0, 256, (B, M, 32), dtype=torch.uint8, device=dev, generator=g)
        td = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=dev, generator=g)
        qxy = (torch.rand((B, M, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev)).contiguous()
        txy = torch.floor(torch.rand((B, N, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev)).contiguous()
        tf = torch.zeros((B, N), dtype=torch.uint8, device=dev)
        out = torch.zeros((B, M, 4), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        # warm-up with a THIRD template instance of the kernel (csr = 2: a 30-px radius), so the rocprofv3 --stats rows of the two
        # reported instances hold 3 untimed + the 35 reported launches each: the clocks settle under the same kind of load
        for _ in range(8):
            lvt.hamming_match_batched(qd, qxy, td, txy, tf, 900.0, 0, H, W, out, launches=10)
        pop = np.array([bin(i).count("1") for i in range(256)], np.int64)

        def sample_check(mode, txy=txy, td=td, N=N):
            """a sample of the big launch against a numpy restatement of the matcher (three problems x their first 48 queries)"""
            checked = 0
            for b in (0, B // 2, B - 1):
                Q = 48
                qd_, td_ = qd[b, :Q].cpu().numpy(), td[b].cpu().numpy()
                qx, tx = qxy[b, :Q].cpu().numpy(), txy[b].cpu().numpy()
                got = out[b, :Q].cpu().numpy()
                d = pop[qd_[:, None, :] ^ td_[None, :, :]].sum(axis=2)
                if mode == 0:   # lvt_image_features_struct.cpp:84-99: squared distance below r^2 (fp32)
                    dx = (tx[None, :, 0] - qx[:, None, 0]).astype(np.float32); dy = (tx[None, :, 1] - qx[:, None, 1]).astype(np.float32)
                    mask = ((dx * dx).astype(np.float32) + (dy * dy).astype(np.float32)).astype(np.float32) < np.float32(625.0)
                else:           # struct.cpp:124-140: rows int(y) - 2 .. int(y) + 2, clipped to [0, rows]
                    qi = qx[:, 1].astype(np.int64)   # the reference compares the train feature's FLOAT y with the integer bounds: kp.y >= start_y && kp.y <= end_y
                    lo, hi = np.maximum(qi - 2, 0).astype(np.float32), np.minimum(qi + 2, H).astype(np.float32)
                    ty = tx[:, 1].astype(np.float32)
                    mask = (ty[None, :] >= lo[:, None]) & (ty[None, :] <= hi[:, None])
                big = np.int64(1) << 40
                key = np.where(mask, d * 65536 + np.arange(N)[None, :], big)
                key = np.concatenate([key, np.full((Q, 2), big)], axis=1)
                o = np.sort(key, axis=1)[:, :2]
                ref = np.stack([np.where(o[:, 0] < big, o[:, 0] % 65536, -1), np.where(o[:, 0] < big, o[:, 0] // 65536, 0x7FFFFFFF),
                                np.where(o[:, 1] < big, o[:, 1] % 65536, -1), np.where(o[:, 1] < big, o[:, 1] // 65536, 0x7FFFFFFF)], axis=1)
                if not np.array_equal(got.astype(np.int64), ref):
                    raise RuntimeError(f"matcher output (mode {mode}) of problem {b} differs from the numpy restatement")
                checked += Q
            return checked

        byts = float(B) * bmatch(M, N)
        lvt.hamming_match_batched(qd, qxy, td, txy, tf, 625.0, 0, H, W, out, launches=3)     # instruction cache, not timed
        us = [lvt.hamming_match_batched(qd, qxy, td, txy, tf, 625.0, 0, H, W, out, launches=5) for _ in range(7)]
        mean = float(np.mean(us))          # = the mean over the 35 reported launches (what tools/profile.sh extracts from the trace)
        ach = byts / (mean * 1e-6) / 1e9
        checked = sample_check(0)
        # the OTHER matcher north_star names: the row-band (left <-> right) instance, measured the same way right behind the radius mode
        for _ in range(8):   # (the sample check above left the GPU idle for a second: settle the clocks again, same third instance)
            lvt.hamming_match_batched(qd, qxy, td, txy, tf, 900.0, 0, H, W, out, launches=10)
        lvt.hamming_match_batched(qd, qxy, td, txy, tf, 0.0, 1, H, W, out, launches=3)
        us_row = [lvt.hamming_match_batched(qd, qxy, td, txy, tf, 0.0, 1, H, W, out, launches=5) for _ in range(7)]
        mean_row = float(np.mean(us_row))
        ach_row = byts / (mean_row * 1e-6) / 1e9
        checked_row = sample_check(1)
        # the row mode's OTHER walk: a train feature whose y is not an in-range integer row (external corners, sub-pixel detectors) sends its problem to the
        # reference's float comparison for every candidate.  Un-floored train coordinates: every problem takes that walk.  Launched at N = 1000 -- template
        # instance <1,1,1,1>, a row of its own in rocprofv3 --stats: the reported instance's row keeps its 3 untimed + 35 timed launches.
        N2 = 1000
        td2 = td[:, :N2].contiguous()
        tf2 = tf[:, :N2].contiguous()
        txy_frac = (torch.rand((B, N2, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev)).contiguous()
        txy_int2 = torch.floor(txy_frac).contiguous()
        byts2 = float(B) * bmatch(M, N2)
        lvt.hamming_match_batched(qd, qxy, td2, txy_int2, tf2, 0.0, 1, H, W, out, launches=3)
        us_int2 = [lvt.hamming_match_batched(qd, qxy, td2, txy_int2, tf2, 0.0, 1, H, W, out, launches=5) for _ in range(3)]
        lvt.hamming_match_batched(qd, qxy, td2, txy_frac, tf2, 0.0, 1, H, W, out, launches=3)
        us_frac = [lvt.hamming_match_batched(qd, qxy, td2, txy_frac, tf2, 0.0, 1, H, W, out, launches=5) for _ in range(3)]
        mean_frac, mean_int2 = float(np.mean(us_frac)), float(np.mean(us_int2))
        checked_frac = sample_check(1, txy_frac, td2, N2)
        del txy_frac, txy_int2, td2, tf2
        # HBM traffic of the same launch from the committed rocprofv3 PMC passes (tools/profile.sh; counters cannot be read from
        # inside this process).  FETCH_SIZE counts the 16-B-per-lane loads of this kernel at one half on gfx950
        # (MI355X_MICROARCH.md, HBM section): corrected bytes = 2 * FETCH_SIZE + WRITE_SIZE.
        traffic, traffic_src = None, None
        live = None if "pmc" in args.skip else live_hamming_traffic(B, M, N)
        if live is not None:
            traffic = round(2.0 * live["FETCH_SIZE"] * 1024.0 + live["WRITE_SIZE"] * 1024.0, 1)
            traffic_src = ("measured by THIS run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, counters only) over "
                           "`python tools/hamming_bench.py %d %d %d 0` = the same launch, mean of %d + %d dispatches: FETCH_SIZE %.1f KB, WRITE_SIZE %.1f KB"
                           % (B, M, N, live["n_FETCH_SIZE"], live["n_WRITE_SIZE"], live["FETCH_SIZE"], live["WRITE_SIZE"]))
        try:
            if traffic is None:
                with open(os.path.join(HERE, "profiles", "hamming_pmc.json")) as f:
                    pm = json.load(f)
                if pm.get("launch") == {"B": B, "M": M, "N": N}:
                    traffic = round(2.0 * pm["FETCH_SIZE_KB_per_launch"] * 1024.0 + pm["WRITE_SIZE_KB_per_launch"] * 1024.0, 1)
                    traffic_src = "profiles/hamming_pmc.json (%s)" % pm.get("source", "rocprofv3 --pmc")
        except Exception:  # noqa: BLE001
            pass
        traffic_row, traffic_row_src = None, None
        live_row = None if "pmc" in args.skip else live_hamming_traffic(B, M, N, mode=1)
        if live_row is not None:
            traffic_row = round(2.0 * live_row["FETCH_SIZE"] * 1024.0 + live_row["WRITE_SIZE"] * 1024.0, 1)
            traffic_row_src = ("measured by THIS run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `python tools/hamming_bench.py %d %d %d 1`, mean of %d + %d "
                               "dispatches: FETCH_SIZE %.1f KB, WRITE_SIZE %.1f KB" % (B, M, N, live_row["n_FETCH_SIZE"], live_row["n_WRITE_SIZE"],
                                                                                         live_row["FETCH_SIZE"], live_row["WRITE_SIZE"]))
        copy_gbs = None
        try:  # what a plain device copy of the same byte count reaches on this box (read half + write half)
            cx = torch.empty(int(byts) // 2, dtype=torch.uint8, device=dev)
            cy = torch.empty_like(cx)
            for _ in range(30):
                cy.copy_(cx)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(10):
                cy.copy_(cx)
            c1.record()
            torch.cuda.synchronize()
            copy_gbs = byts / (c0.elapsed_time(c1) * 1e-4) / 1e9
            del cx, cy
        except Exception:  # noqa: BLE001
            pass
        return {"roofline": {
            "kernel": "lvt::k_hamming_batched<0,3,1,2> (masked 2-NN Hamming matcher, radius mode)", "bound": "hbm", "achieved": round(ach, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_us": round(mean, 2),
            "median_us": round(float(np.median(us)), 2), "launch": {"B": B, "M": M, "N": N, "algorithmic_bytes": byts},
            "traffic_source": traffic_src, "output_checked": f"{checked} queries of 3 problems == numpy restatement",
            "device_copy_same_bytes_GBs": None if copy_gbs is None else round(copy_gbs, 1),
            "frac_of_device_copy": None if copy_gbs is None else round(ach / copy_gbs, 4),
            "row_mode": {"kernel": "lvt::k_hamming_batched<1,1,1,2> (the same matcher with the row-band mask of row_match, lvt_image_features_struct.cpp:122-148: "
                                   "the stereo left <-> right instance)", "bound": "hbm", "achieved": round(ach_row, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach_row / HBM_PEAK_GBS, 4), "traffic": traffic_row, "avg_us": round(mean_row, 2), "median_us": round(float(np.median(us_row)), 2),
                         "launch": {"B": B, "M": M, "N": N, "algorithmic_bytes": byts}, "traffic_source": traffic_row_src,
                         "output_checked": f"{checked_row} queries of 3 problems == numpy restatement",
                         "fractional_rows": {"what": "instance <1,1,1,1> (B x 1000 x 1000) with FRACTIONAL train rows (external corners): every problem takes the walk that re-checks "
                                                     "kp.y >= start_y && kp.y <= end_y per candidate, against the same launch with integer rows (every detector output: the lean walk)",
                                             "launch": {"B": B, "M": M, "N": N2, "algorithmic_bytes": byts2},
                                             "avg_us": round(mean_frac, 2), "frac": round(byts2 / (mean_frac * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                             "integer_rows_avg_us": round(mean_int2, 2), "integer_rows_frac": round(byts2 / (mean_int2 * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                             "output_checked": f"{checked_frac} queries of 3 problems == numpy restatement (float row compare)"}},
            "note": "B independent KITTI-nominal problems per launch; algorithmic bytes 40(M+N)+N+16M each (SURVEY 8d); MEAN of 35 launches "
                    "(7 x 5 back to back between two HIP events on the launch stream) after 80 warm-up launches of a THIRD template instance "
                    "(csr = 2) in front of each of the two reported instances (radius mode first, then `row_mode`); the rocprofv3 --stats average of this kernel over the same command is the same statistic (profiles/); "
                    "traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950 counts this kernel's 16-B-per-lane loads at one half) of the same launch: "
                    "collected by this run through rocprofv3 when the tool is there and this process is not itself being profiled, else the "
                    "committed passes of tools/profile.sh (traffic_source says which).  Ceiling of this access pattern on an MI355X, measured in round 4 "
                    "(profiles/r04_hamming_analysis.md 3b): the launch's loads and stores ALONE take 174 us = 0.69 of the 8 TB/s spec"}}

    def _leg_sync(self, args):
        """one frame at a time, nothing overlapped: what a caller of the reference's lvt_track gets (lvt_c.cpp:63-88); mean of the
        wall time of a synchronous call is the reference's own metric (kitti_example.cpp:129-131,143-149)"""
        torch, lvt = self.torch, self.lvt
        n = max(args.sync_frames, 110)   # always >= 100 measured calls, whatever --steps / --warmup are: the leg renders its own frames
        H, W, pitch = self.H, self.W, self.pitch
        if n <= self.n_frames:
            dev_frames = self.frames[0, :n]
        else:
            dev_frames = torch.zeros((n, 2, H, pitch), dtype=torch.uint8, device=self.device)
            dev_frames[:self.n_frames] = self.frames[0]
            for i in range(self.n_frames, n):
                dev_frames[i, :, :, :W] = self.worlds[0].render_stereo_torch(i, device=self.device)
            self.sync()
        host = dev_frames[:, :, :, :W].contiguous().cpu()
        pinned = host.pin_memory()
        host_np, pinned_np = host.numpy(), pinned.numpy()
        res = {}
        for name in ("lvt_track_host", "lvt_track_pinned", "track_device"):
            vo = lvt.LvtSystem.create(self.prm, 1)
            ts = []
            for i in range(n):
                t0 = time.perf_counter()
                if name == "lvt_track_host":
                    vo.track(host_np[i, 0], host_np[i, 1])
                elif name == "lvt_track_pinned":
                    vo.track(pinned_np[i, 0], pinned_np[i, 1])
                else:
                    l = dev_frames[i].data_ptr()
                    vo.track_device(l, l + H * pitch, H, W, pitch)
                ts.append(1e3 * (time.perf_counter() - t0))
            ok = vo.get_state() == 2 and vo.last_error() == ""
            hs = vo.host_stats()
            res[name + "_route"] = {"planes_read_in_place": hs["planes_in_place"], "planes_copied_to_staging": hs["planes_staged"]}
            ts = ts[10:]
            res[name + "_ms"] = {"p50": round(pct(ts, 50), 4), "p99": round(pct(ts, 99), 4), "mean": round(float(np.mean(ts)), 4), "tracking": ok}
            vo.close()
        res["frames"] = n - 10
        res["fps_lvt_track_host_mean"] = round(1e3 / res["lvt_track_host_ms"]["mean"], 1)
        res["note"] = ("lvt_track = the reference's C-ABI call with borrowed pageable host buffers (CPU copy into a pinned staging buffer + one pull "
                       "kernel); pinned = the caller's buffers are page-locked and the pull kernel reads them where they lie (the *_route counters "
                       "say which way every plane went); track_device = images already in HBM")
        return {"sync": res}

    def _batch_run(self, fr, S, n, H, W, pitch, profile=False, depth=3):
        """S sequences in lock-step over frames [0, n) of fr[:S]; returns (seconds of the timed part, frames not TRACKING, error, handle's
        per-kernel profile or None, counts of the last frame per sequence)"""
        vo = self.lvt.LvtBatch(self.prm, S)
        lp = [[fr[s, i, 0].data_ptr() for s in range(S)] for i in range(n)]
        rp = [[fr[s, i, 1].data_ptr() for s in range(S)] for i in range(n)]
        wm = 4
        for i in range(wm):
            vo.track_device_async(lp[i], rp[i], H, W, pitch); vo.wait()
        if profile:
            vo.profile_enable(True)
        self.sync()
        t0 = time.perf_counter()
        inflight, bad = 0, 0
        depth = 1 if profile else depth   # lock-step frames in flight (3: the feature stage of frame t+2 is already queued when frame t ends)
        for i in range(wm, n):
            vo.track_device_async(lp[i], rp[i], H, W, pitch); inflight += 1
            if inflight >= depth:
                bad += int((vo.wait()[2] != 2).sum()); inflight -= 1
        while inflight:
            bad += int((vo.wait()[2] != 2).sum()); inflight -= 1
        self.sync()
        dt = time.perf_counter() - t0
        prof = vo.profile_read() if profile else None
        cnt = [vo.counts(s) for s in range(S)]
        err = vo.last_error()
        self._last_score_pieces = vo.host_stats()["score_pieces"]   # launches the batch's k_score went out in at the end of the run (1 / 2: chosen at run time)
        vo.close()
        return dt, bad, err, prof, cnt

    def _leg_batch(self, args):
        """S sequences in lock-step on this GPU (lvt_amd_batch_*): the headline batch size, a sweep over batch sizes, and the pipeline's
        OWN matcher launches (k_hamming_batched_lists, the list-emitting form of the binned matcher) against the HBM roof at 64 sequences"""
        S0, n = args.batch_seqs, args.batch_frames
        if S0 <= 1 or n <= 4:
            return {}
        sweep_sizes = sorted(set([12, 16, 24, 32, 48, S0]))
        Smax = max(sweep_sizes + [64])
        worlds = [self.make_world("kitti", seed=100 + s) for s in range(Smax)]
        fr, H, W, pitch = self._render(worlds, n)
        out = {}
        sweep = []
        for S in sweep_sizes:
            dt, bad, err, _, _ = self._batch_run(fr, S, n, H, W, pitch, depth=args.batch_depth)
            row = {"seqs": S, "frames_each": n - 4, "frames_in_flight": args.batch_depth, "fps": round(S * (n - 4) / dt, 1), "ms_per_lockstep_frame": round(1e3 * dt / (n - 4), 4),
                   "frames_not_tracking": bad, "error": err, "score_pieces_at_end": self._last_score_pieces}
            sweep.append(row)
            if S == S0:
                out["batch"] = row
        out["batch_sweep"] = sweep
        # the matcher ON THE PIPELINE'S PATH: per-launch HIP-event time of the list kernels in a 64-sequence lock-step batch (profiling
        # serialises the frame: one frame in flight), algorithmic bytes from the per-sequence counts of the last frame
        m = min(n, 16)
        dt, bad, err, prof, cnt = self._batch_run(fr, 64, m, H, W, pitch, profile=True)
        rows = {}
        for name, ms, calls in prof:
            if "k_hamming_batched_lists" in name and calls:
                rows[name] = 1e3 * ms / calls
        b_map = float(sum(bmatch(c["map_size_at_match"], c["n_left"]) for c in cnt))
        b_row = float(sum(bmatch(c["n_left"], c["n_right"]) for c in cnt))
        pm = {}
        for name, us in rows.items():
            byts = b_row if "(row)" in name else b_map
            ach = byts / (us * 1e-6) / 1e9
            pm[name] = {"avg_us": round(us, 2), "algorithmic_bytes_per_launch": round(byts), "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 5)}
        if pm:
            dom = max(pm, key=lambda k: pm[k]["avg_us"])
            out["roofline_pipeline_matcher"] = {
                "kernel": "lvt::k_hamming_batched_lists<2 (row) / 0 (map), false> -- what a lock-step batch launches for find_match_index / row_match",
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "sequences": 64, "dominant": dom, "achieved": pm[dom]["achieved"],
                "frac": pm[dom]["frac"], "traffic": None, "launches": pm, "frames_not_tracking": bad, "error": err,
                "note": "64 KITTI-shaped problems per launch (one workgroup per sequence, two for the row lists): 64 x ~100 KB is latency-bound "
                        "work on 64 / 128 of 256 CUs -- the roofline fraction of the batched matcher proper (`roofline`) needs thousands of problems"}
        del fr
        return out

    def _leg_lists_ab(self, args):
        """single handle, asynchronous pipeline: the wave-per-query list kernels (default) against the binned list kernel forced onto
        the handle (LVT_AMD_BINNED_LISTS=1, read at creation) on the same frames"""
        n = min(self.n_frames, 220)
        if n < 40:
            return {}
        res = {}
        for tag, env in (("wave_per_query", "0"), ("binned_lists", "1")):
            os.environ["LVT_AMD_BINNED_LISTS"] = env
            try:
                vo = self.lvt.LvtSystem.create(self.prm, 1)
            finally:
                os.environ.pop("LVT_AMD_BINNED_LISTS", None)
            for i in range(10):
                l, r = self._ptrs(i)
                vo.track_device(l, r, self.H, self.W, self.pitch)
            self.sync()
            t0 = time.perf_counter()
            inflight, bad = 0, 0
            for i in range(10, n):
                l, r = self._ptrs(i)
                vo.track_device_async(l, r, self.H, self.W, self.pitch); inflight += 1
                if inflight >= 4:
                    bad += 0 if vo.wait_status()[2] == 2 else 1; inflight -= 1
            while inflight:
                bad += 0 if vo.wait_status()[2] == 2 else 1; inflight -= 1
            dt = time.perf_counter() - t0
            res[tag] = {"fps": round((n - 10) / dt, 1), "frames_not_tracking": bad}
            vo.close()
        return {"single_handle_list_kernels": res}

    def _async_fps(self, vo, frames, depth=4, warm=6):
        """frames/s of the asynchronous host-buffer entry (lvt_amd_track_async / lvt_amd_track_rgbd_async + lvt_amd_wait_status) over page-locked
        frames, `depth` frames in flight, the first `warm` frames outside the window; returns (fps, frames not TRACKING, frames timed)"""
        inflight, bad = 0, 0
        for a, b in frames[:warm]:
            assert vo.track_async(a, b) == 0
            inflight += 1
            if inflight >= depth:
                vo.wait(); inflight -= 1
        while inflight:
            vo.wait(); inflight -= 1
        self.sync()
        t0 = time.perf_counter()
        for a, b in frames[warm:]:
            vo.track_async(a, b); inflight += 1
            if inflight >= depth:
                bad += 0 if vo.wait_status()[2] == 2 else 1; inflight -= 1
        while inflight:
            bad += 0 if vo.wait_status()[2] == 2 else 1; inflight -= 1
        self.sync()
        dt = time.perf_counter() - t0
        return (len(frames) - warm) / dt, bad, len(frames) - warm

    def _leg_configs(self, args):
        """BASELINE.json configs[2] / configs[3] shapes, synchronous calls (ms per frame), every frame checked TRACKING"""
        lvt = self.lvt
        res = {}
        n = args.config_frames
        if n <= 6:
            return {}
        # EuRoC-shaped 752x480 stereo, device-resident
        w = self.make_world("euroc", seed=0)
        fr, H, W, pitch = self._render([w], n)
        vo = lvt.LvtSystem.create(lvt.euroc_params(), 1)
        ts, bad = [], 0
        for i in range(n):
            p = fr[0, i].data_ptr()
            t0 = time.perf_counter()
            vo.track_device(p, p + H * pitch, H, W, pitch)
            ts.append(1e3 * (time.perf_counter() - t0))
            bad += 0 if vo.get_state() == 2 else 1
        c = vo.counts()
        res["euroc_752x480_stereo"] = {"ms_per_frame_p50": round(pct(ts[5:], 50), 4), "fps_sync": round(1e3 / float(np.mean(ts[5:])), 1), "frames": n - 5,
                                       "frames_not_tracking": bad, "features_left": c["n_left"], "map_size": c["map_size"], "entry": "lvt_amd_track_device"}
        vo.close()
        # the metric itself (frames/s) on this shape: asynchronous host frames, page-locked, tightly packed -- the entry the headline uses
        torch = self.torch
        hostf = fr[0, :, :, :, :W].contiguous().cpu()
        pin = hostf.pin_memory().numpy()
        vo = lvt.LvtSystem.create(lvt.euroc_params(), 1)
        fps, bad, nt = self._async_fps(vo, [(pin[i, 0], pin[i, 1]) for i in range(n)])
        res["euroc_752x480_stereo"].update({"fps_async": round(fps, 1), "fps_async_frames": nt, "fps_async_frames_not_tracking": bad,
                                            "fps_async_entry": "lvt_amd_track_async (page-locked host frames, 4 in flight) + lvt_amd_wait_status"})
        vo.close()
        del fr
        # TUM-shaped 640x480 RGB-D (host buffers: the RGB-D entry point takes gray u8 + depth f32 like lvt_system::track)
        w = self.make_world("tum", seed=0)
        m = min(n, 40)
        frames = [w.render_rgbd(i) for i in range(m)]
        vo = lvt.LvtSystem.create(lvt.tum_params(), 2)
        ts, bad = [], 0
        for a, b in frames:
            t0 = time.perf_counter()
            vo.track(a, b)
            ts.append(1e3 * (time.perf_counter() - t0))
            bad += 0 if vo.get_state() == 2 else 1
        c = vo.counts()
        res["tum_640x480_rgbd"] = {"ms_per_frame_p50": round(pct(ts[5:], 50), 4), "fps_sync": round(1e3 / float(np.mean(ts[5:])), 1), "frames": m - 5,
                                   "frames_not_tracking": bad, "features_left": c["n_left"], "map_size": c["map_size"], "entry": "lvt_amd_track_rgbd (host buffers)"}
        vo.close()
        # the same frames from page-locked buffers (read in place: no CPU copy of the 1.2-MB depth image into the staging buffer)
        pinned = [(torch.from_numpy(a).pin_memory().numpy(), torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).pin_memory().numpy()) for a, b in frames]
        vo = lvt.LvtSystem.create(lvt.tum_params(), 2)
        ts, bad = [], 0
        for a, b in pinned:
            t0 = time.perf_counter()
            vo.track(a, b)
            ts.append(1e3 * (time.perf_counter() - t0))
            bad += 0 if vo.get_state() == 2 else 1
        res["tum_640x480_rgbd"]["ms_per_frame_p50_pinned_buffers"] = round(pct(ts[5:], 50), 4)
        res["tum_640x480_rgbd"]["frames_not_tracking"] += bad
        vo.close()
        vo = lvt.LvtSystem.create(lvt.tum_params(), 2)
        fps, bad, nt = self._async_fps(vo, pinned)
        res["tum_640x480_rgbd"].update({"fps_async": round(fps, 1), "fps_async_frames": nt, "fps_async_frames_not_tracking": bad,
                                        "fps_async_entry": "lvt_amd_track_rgbd_async (page-locked gray + depth, 4 in flight) + lvt_amd_wait_status"})
        vo.close()
        return {"configs": res}

    def _leg_cpu(self, args, warm, poses):
        """the CPU oracle (port of the reference path, 2 threads like the reference) on the same frames: baseline + per-frame SE3"""
        from oracle import pyoracle as O
        nf = min(self.n_frames, args.cpu_frames)
        host = self.frames[0, :nf, :, :, :self.W].contiguous().cpu().numpy()
        orc = O.Oracle(self.prm, 1, threads=2)
        gpu = list(warm) + list(poses)
        own = len(gpu)
        # a short run (the driver's --steps 20 --warmup 5) still holds >= SE3_MIN_FRAMES poses against the oracle: the sequence is tracked AGAIN from frame 0
        # on a fresh handle through the same asynchronous entry, outside the timed window, the frames past the run's own rendered here; its first poses must
        # BE the run's own (same frames, same entry: bit for bit), the rest extends the comparison
        SE3_MIN_FRAMES = 200
        replay_identical, replay_close = None, True
        if nf < SE3_MIN_FRAMES and args.cpu_frames >= SE3_MIN_FRAMES:
            extra = np.stack([self.worlds[0].render_stereo_torch(i, device=self.device).cpu().numpy() for i in range(nf, SE3_MIN_FRAMES)])
            host = np.ascontiguousarray(np.concatenate([host, extra], axis=0))
            nf = SE3_MIN_FRAMES
            vo2 = self.lvt.LvtSystem.create(self.prm, 1)
            rep, inflight = [], 0
            for i in range(nf):
                vo2.track_async(host[i, 0], host[i, 1])
                inflight += 1
                if inflight >= 4:
                    rep.append(vo2.wait_status()); inflight -= 1
            while inflight:
                rep.append(vo2.wait_status()); inflight -= 1
            vo2.close()
            replay_identical = all(np.array_equal(np.asarray(gpu[i][0]), rep[i][0]) and np.array_equal(np.asarray(gpu[i][1]), rep[i][1]) for i in range(own))
            replay_close = all(np.allclose(np.asarray(gpu[i][0]), rep[i][0], atol=1e-9, rtol=0) and np.allclose(np.asarray(gpu[i][1]), rep[i][1], atol=1e-9, rtol=0) for i in range(own))
            gpu = gpu + [(r[0], r[1]) for r in rep[own:]]
        tc = time.perf_counter()
        done, max_et, max_er, worst = 0, 0.0, 0.0, -1
        for i in range(nf):
            Ro, to = orc.track(host[i, 0], host[i, 1])
            done += 1
            Rh, th = gpu[i][0], gpu[i][1]
            e_t, e_R = pose_errors(np.asarray(Rh), np.asarray(th), Ro, to)
            if max(e_t, e_R) > max(max_et, max_er):
                worst = i
            max_et, max_er = max(max_et, e_t), max(max_er, e_R)
            if time.perf_counter() - tc > args.cpu_seconds:
                break
        tcpu = time.perf_counter() - tc
        out = {"se3": {"frames": done, "max_e_t": max_et, "max_e_R_rad": max_er, "tol": POSE_TOL, "worst_frame": worst,
                       "pass": bool(max_et <= POSE_TOL and max_er <= POSE_TOL and orc.status == 2 and replay_close),
                       "frames_of_the_run_itself": min(own, done), "replay_identical_to_the_run": replay_identical,
                       "reference": "oracle/liblvt_oracle.so (CPU restatement of the reference path; parity unpinned, see DESIGN.md section 5) on "
                                    "the same frames: warm-up and timed frames through the asynchronous entry the headline uses"},
               "cpu_baseline": {"value": round(done / tcpu, 2), "unit": "frames/s", "cores": 2, "kind": "port",
                                "sample": f"first {done} stereo pairs of the rank-0 sequence, oracle/liblvt_oracle.so with the reference's 2-thread "
                                          f"left/right split (the pose comparison runs inside this loop); host has {os.cpu_count()} logical cores"}}
        try:  # 8 independent sequences side by side, 2 threads each (ctypes releases the GIL inside the oracle)
            import threading
            n_par = min(8, max(1, (os.cpu_count() or 2) // 2))
            npf = min(nf, 120)
            orcs = [O.Oracle(self.prm, 1, threads=2) for _ in range(n_par)]

            def _run(o):
                for i in range(npf):
                    o.track(host[i, 0], host[i, 1])
            ths = [threading.Thread(target=_run, args=(o,)) for o in orcs]
            tp = time.perf_counter()
            for t_ in ths:
                t_.start()
            for t_ in ths:
                t_.join()
            tpar = time.perf_counter() - tp
            out["cpu_baseline"]["parallel"] = {"sequences": n_par, "cores": 2 * n_par, "value": round(n_par * npf / tpar, 2), "unit": "frames/s",
                                               "sample": f"{n_par} oracle instances x the first {npf} stereo pairs, concurrently"}
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"]["parallel"] = {"error": str(e)}
        return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--depth", type=int, default=0, help="poses outstanding in the async pipeline (1 = synchronous; default: 4 for one sequence per GPU, 3 for a lock-step batch)")
    ap.add_argument("--seqs-per-gpu", type=int, default=1, help="independent sequences advanced in lock-step on each GPU (cfg 5 on G < 8 GPUs: ceil(8 / G))")
    ap.add_argument("--profile-steps", type=int, default=40, help="extra frames run with per-kernel HIP events")
    ap.add_argument("--cpu-frames", type=int, default=600, help="upper bound of frames run through the CPU oracle (baseline + SE3 check)")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--hamming-batch", type=int, default=8192, help="problems per matcher launch (SURVEY 8d: the roofline fraction is reported on the largest batch)")
    ap.add_argument("--sync-frames", type=int, default=160)
    ap.add_argument("--batch-seqs", type=int, default=16)
    ap.add_argument("--batch-frames", type=int, default=44)
    ap.add_argument("--batch-depth", type=int, default=3, help="lock-step frames in flight in the batch leg")
    ap.add_argument("--config-frames", type=int, default=60)
    ap.add_argument("--total-seqs", type=int, default=0, help="track this many sequences in all: ceil(T / gpus) per GPU in lock-step (cfg 5 with T = 8 on fewer than 8 GPUs)")
    ap.add_argument("--device-frames", action="store_true", help="headline on frames already resident in HBM (lvt_amd_track_device_async) instead of pinned host frames")
    ap.add_argument("--share-gpu", action="store_true", help="ranks beyond the box's GPU count share its GPUs (local_rank mod device count): rehearses the multi-rank "
                    "launch, timing and reduction path on a one-GPU box; the number it prints is not a scaling result")
    ap.add_argument("--backend", default="hip", choices=["hip", "standin"], help="standin: sleeps instead of GPU work (CPU test of the multi-rank launch path)")
    ap.add_argument("--standin-ms", type=float, default=0.0, help="stand-in backend: every rank's step takes this long (default: rank r takes 1 + r ms)")
    ap.add_argument("--standin-tail-ms", type=float, default=0.0, help="stand-in backend: rank 1 sleeps this long between its timed window and the reductions")
    ap.add_argument("--skip", default="", help="comma-separated side legs to skip: kernels,roofline,pmc (the live counter passes of the roofline leg),sync,batch,lists_ab,configs,cpu")
    args = ap.parse_args(argv)
    args.skip = [s for s in args.skip.split(",") if s]
    args.depth_auto = args.depth <= 0
    if args.depth_auto:
        args.depth = 4 if args.seqs_per_gpu == 1 else 3
    return args


class StandInBackend:
    """--backend standin: the five methods run_rank needs, with sleeps for steps (no GPU).  It exists so that the N > 1 launch path --
    self-launch, rendezvous, barriers, reductions, the one JSON line -- runs end to end on a CPU box (tests/test_shard_gloo.py);
    rank r's step takes (1 + r) ms, rank 1 reports one lost frame."""
    device = None
    workload = "stand-in (sleeps; no GPU work)"

    def __init__(self, args, env):
        self.env = env
        self.args = args
        self.sequences_per_gpu = args.seqs_per_gpu

    def sync(self):
        pass

    def prepare(self, n_frames):
        self.n_frames = n_frames

    def warmup(self, Wm):
        return [None] * Wm

    def timed(self, first, K, depth):
        ms = self.args.standin_ms if self.args.standin_ms > 0 else (1 + self.env.rank)
        for _ in range(K):
            time.sleep(0.001 * ms)
        return [(None, None)] * K, (1 if self.env.rank == 1 else 0)

    def tail(self):
        """(tests) a rank that dawdles between its timed window and the reductions: must not show in anybody's time"""
        if self.args.standin_tail_ms > 0 and self.env.rank == 1:
            time.sleep(0.001 * self.args.standin_tail_ms)

    def extras(self, args, env, warm, poses):
        return {"roofline": None, "cpu_baseline": None}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` from a plain shell: start the N ranks the way the driver does (one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1) and hand their exit code on; rank 0 prints the one JSON line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL would fail without it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    env = rank_env()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if env.world_size != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {env.world_size}")
    if args.total_seqs > 0:  # cfg 5 on G < 8 GPUs: ceil(total / G) sequences per GPU in lock-step
        args.seqs_per_gpu = -(-args.total_seqs // env.world_size)
        args.effective_total_seqs = args.seqs_per_gpu * env.world_size   # (rounded UP when total % gpus != 0: reported in config)
        if args.depth_auto:
            args.depth = 4 if args.seqs_per_gpu == 1 else 3
    standin = args.backend == "standin"
    backend = StandInBackend(args, env) if standin else HipBackend(args, env)
    dist = None
    if env.world_size > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        # gloo for every backend: the ranks exchange one barrier and a few scalars OUTSIDE the timed window, there is no data-path collective
        #  -- bringing RCCL up (IPC handles, xGMI topology) would only add a way to fail
        dist.init_process_group("gloo")
    result = run_rank(args, env, dist, backend)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if env.rank == 0:
        print(json.dumps(result), flush=True)
        se3 = result.get("se3")
        if not standin and (result["tracking"]["frames_not_tracking"] or (se3 is not None and not se3["pass"])):
            raise SystemExit("bench.py: the run left TRACKING or its poses differ from the CPU reference by more than 1e-4 -- the number above is INVALID")


if __name__ == "__main__":
    main()
