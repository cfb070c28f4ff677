#!/usr/bin/env python3
"""bench.py -- stereo frames/sec of the MI355X-native LVT tracking path on KITTI-shaped input.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU).
A "step" is one pass of the hot path over one stereo pair (one lvt_track call) of a synthetic KITTI-00-shaped
sequence (1241x376, BASELINE.json configs[1]); independent sequences (seed = rank) shard one per GPU, there is
no collective on the data path -> "scaling": "weak".  Frames are pre-rendered into HBM before the
timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s is achievable


def bmatch(M, N):
    """algorithmic bytes of one masked-Hamming problem: 40(M+N) + N + 16M"""
    return 40 * (M + N) + N + 16 * M


def bench_batched(args, torch, lvt_amd, make_world, dist, dev, rank, world_size):
    """S independent sequences per GPU in lock-step (lvt_amd_batch_*): same metric, more sequences per device."""
    K, Wm, S = args.steps, args.warmup, args.seqs_per_gpu
    worlds = [make_world("kitti", seed=rank * S + s) for s in range(S)]
    prm = lvt_amd.kitti_params()
    H, W = worlds[0].H, worlds[0].W
    pitch = ((W + 63) // 64) * 64
    n_frames = Wm + K
    frames = torch.zeros((S, n_frames, 2, H, pitch), dtype=torch.uint8, device=dev)
    for s in range(S):
        for i in range(n_frames):
            frames[s, i, :, :, :W] = worlds[s].render_stereo_torch(i, device=dev)
    torch.cuda.synchronize()
    vo = lvt_amd.LvtBatch(prm, S)
    lp = [[frames[s, i, 0].data_ptr() for s in range(S)] for i in range(n_frames)]
    rp = [[frames[s, i, 1].data_ptr() for s in range(S)] for i in range(n_frames)]
    for i in range(Wm):
        vo.track_device_async(lp[i], rp[i], H, W, pitch); vo.wait()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    t0 = time.perf_counter()
    inflight = 0
    last = None
    for i in range(Wm, Wm + K):
        vo.track_device_async(lp[i], rp[i], H, W, pitch)
        inflight += 1
        if inflight >= args.depth:
            last = vo.wait(); inflight -= 1
    while inflight:
        last = vo.wait(); inflight -= 1
    torch.cuda.synchronize()
    if dist: dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    n_lost = int((last[2] != 2).sum())
    if dist:
        dist.barrier(); dist.destroy_process_group()
    if rank == 0:
        fps = world_size * S * K / elapsed
        print(json.dumps({
            "metric": "stereo frames/sec (KITTI-shaped 1241x376), per-frame SE3 vs CPU ref", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world_size, "steps": K, "warmup": Wm, "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/f64", "data": "synthetic",
            "config": {"workload": f"{S} independent KITTI seq 00-shaped synthetic stereo sequences per GPU advanced in lock-step "
                                   "(one step = one stereo pair of EVERY sequence), frames resident in HBM",
                       "sequences_per_gpu": S, "frames_in_flight": args.depth, "parallelism": f"{world_size * S} independent sequences, no collective"},
            "tracking": {"lost_sequences": n_lost, "error": vo.last_error()}, "roofline": None, "cpu_baseline": None}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--profile-steps", type=int, default=100, help="extra frames run with per-kernel HIP events")
    ap.add_argument("--cpu-frames", type=int, default=600, help="upper bound of frames timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--depth", type=int, default=4, help="poses outstanding in the async pipeline (1 = synchronous)")
    ap.add_argument("--hamming-batch", type=int, default=8192, help="problems per matcher launch (SURVEY 8d: the roofline fraction is reported on the largest batch)")
    ap.add_argument("--seqs-per-gpu", type=int, default=1, help="independent sequences advanced in lock-step on each GPU")
    args = ap.parse_args()

    import torch
    import lvt_amd
    from lvt_amd.synth import make_world

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        if args.gpus > 1 and world_size == 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the tracking path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world_size > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", device_id=dev)

    K, Wm, P = args.steps, args.warmup, args.profile_steps
    S = args.seqs_per_gpu
    if S > 1:
        return bench_batched(args, torch, lvt_amd, make_world, dist, dev, rank, world_size)
    world = make_world("kitti", seed=rank)
    prm = lvt_amd.kitti_params()
    H, W = world.H, world.W
    pitch = ((W + 63) // 64) * 64
    n_frames = Wm + K + P
    # ---- synthetic frames rendered straight into HBM (pitched, zero padded)
    frames = torch.zeros((n_frames, 2, H, pitch), dtype=torch.uint8, device=dev)
    for i in range(n_frames):
        frames[i, :, :, :W] = world.render_stereo_torch(i, device=dev)
    torch.cuda.synchronize()
    base = frames.data_ptr()
    fstride = 2 * H * pitch

    vo = lvt_amd.LvtSystem.create(prm, lvt_amd.eSensor_STEREO)

    def run(i):
        p = base + i * fstride
        return vo.track_device(p, p + H * pitch, H, W, pitch)

    def run_async(i):
        p = base + i * fstride
        vo.track_device_async(p, p + H * pitch, H, W, pitch)

    # frames are enqueued asynchronously with at most DEPTH poses outstanding: the feature stage of frame t+1
    # overlaps the tracking chain of frame t on the device, the host never idles the GPU between frames
    DEPTH = args.depth
    n_lost = 0
    for i in range(Wm):
        run(i)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    t0 = time.perf_counter()
    inflight = 0
    poses = []
    for i in range(Wm, Wm + K):
        run_async(i)
        inflight += 1
        if inflight >= DEPTH:
            poses.append(vo.wait()); inflight -= 1
    while inflight:
        poses.append(vo.wait()); inflight -= 1
    torch.cuda.synchronize()
    if dist: dist.barrier()
    elapsed = time.perf_counter() - t0
    assert len(poses) == K
    n_lost = 0 if vo.get_state() == lvt_amd.eState_TRACKING else 1   # LOST is sticky: the final state tells
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        lost = torch.tensor([n_lost], dtype=torch.int64, device=dev)
        dist.all_reduce(lost, op=dist.ReduceOp.SUM)
        n_lost = int(lost.item())
    counts = vo.counts()
    err = vo.last_error()

    result = None
    if rank == 0:
        # ---- per-kernel timing pass (HIP events on the launch stream) over the next P frames
        kernels = []
        roofline = None
        roofline_dom = None
        frame_bytes = None
        if P > 0:
            vo.profile_enable(True)
            nl = nr = mp = 0
            for i in range(Wm + K, Wm + K + P):
                run(i)
                c = vo.counts()
                nl += c["n_left"]; nr += c["n_right"]; mp += c["map_size_at_match"]
            prof = vo.profile_read()
            vo.profile_enable(False)
            nl /= P; nr /= P; mp /= P
            alg = {  # algorithmic HBM bytes per launch (DESIGN.md "roofline accounting")
                "k_score": 2.0 * W * H,
                "k_cells(pass0)": 0.0, "k_gather": 0.0,
                "k_brief": (nl + nr) * 40.0,
                "k_candidates(map)": bmatch(mp, nl), "k_resolve(map)": 16.0 * mp + nl,
                "k_candidates(row)": bmatch(nl, nr), "k_resolve(row)": 16.0 * nl + nr,
                "k_candidates(staged)": 0.0,
                "k_pnp": counts["n_matches"] * 32.0 + 56.0,
            }
            # (the gate kernels -- and k_match_map's head for a single sequence -- wait for another stream: their event time is mostly waiting)
            work = [x for x in prof if "k_gate" not in x[0] and "wait for" not in x[0]]
            tot = sum(ms for _, ms, _ in work)
            for name, ms, calls in prof:
                if calls:
                    gate = "k_gate" in name or "wait for" in name
                    kernels.append({"kernel": name, "avg_us": round(1e3 * ms / calls, 3), "share": None if gate else round(ms / tot, 4)})
            # the frame period is the tracking stream's chain (the feature and early streams run beside it): its longest kernel
            chain = [x for x in work if x[0].startswith(("k_match_map", "k_track_mid", "k_pnp", "k_candidates(staged)", "k_triangulate"))]
            dom = max(chain or work, key=lambda x: x[1])
            dom_us = 1e3 * dom[1] / max(dom[2], 1)
            ab = next((v for k, v in alg.items() if dom[0].startswith(k)), 0.0)
            ach = ab / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
            roofline_dom = {"kernel": dom[0], "bound": "fp64 VALU issue of one CU + serial 6x6 solve (Levenberg-Marquardt), not HBM", "achieved": round(ach, 4),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 8), "traffic": None,
                            "avg_us": round(dom_us, 3), "algorithmic_bytes_per_launch": round(ab, 1)}
            frame_bytes = 2.0 * W * H + (nl + nr) * 40.0 + bmatch(mp, nl) + bmatch(nl, nr) + 24.0 * mp + 56.0
        # ---- batched Hamming matcher micro-benchmark (the kernel north_star prices against the HBM roof)
        hb = None
        try:
            B, M, N = args.hamming_batch, 1000, 1500
            g = torch.Generator(device=dev); g.manual_seed(1234)
            qd = torch.randint(0, 256, (B, M, 32), dtype=torch.uint8, device=dev, generator=g)
            td = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=dev, generator=g)
            qxy = torch.rand((B, M, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev)
            txy = torch.floor(torch.rand((B, N, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev))
            tf = torch.zeros((B, N), dtype=torch.uint8, device=dev)
            out = torch.zeros((B, M, 4), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            qxy, txy = qxy.contiguous(), txy.contiguous()
            # warm-up: the launch time keeps falling for the first ~50 launches after the light pipeline phase (270 -> 229 us
            # measured) while the clocks ramp up under the sustained load; the steady state is what is reported
            for _ in range(8):
                lvt_amd.hamming_match_batched(qd, qxy, td, txy, tf, 625.0, 0, H, W, out, launches=10)
            us = []
            for _ in range(7):  # 5 launches back to back per timing: the average excludes the ~5 us of launch latency
                us.append(lvt_amd.hamming_match_batched(qd, qxy, td, txy, tf, 625.0, 0, H, W, out, launches=5))
            us = sorted(us)
            med = us[len(us) // 2]
            byts = float(B) * bmatch(M, N)
            ach = byts / (med * 1e-6) / 1e9
            # HBM traffic of the same launch from the rocprofv3 PMC passes (tools/profile.sh; counters cannot be read from
            # inside this process).  FETCH_SIZE counts the 16-B-per-lane loads of this kernel at one half on gfx950
            # (MI355X_MICROARCH.md, HBM section): corrected bytes = 2 * FETCH_SIZE + WRITE_SIZE.
            traffic, traffic_src = None, None
            try:
                pj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hamming_pmc.json")
                with open(pj) as f:
                    pm = json.load(f)
                if pm.get("launch") == {"B": B, "M": M, "N": N}:
                    traffic = round(2.0 * pm["FETCH_SIZE_KB_per_launch"] * 1024.0 + pm["WRITE_SIZE_KB_per_launch"] * 1024.0, 1)
                    traffic_src = "profiles/hamming_pmc.json (%s)" % pm.get("source", "rocprofv3 --pmc")
            except Exception:  # noqa: BLE001
                pass
            # what a plain device copy of the same byte count reaches on this box (read half + write half): the achievable
            # streaming rate, reported beside the 8 TB/s spec peak that `frac` is priced against
            copy_gbs = None
            try:
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
            hb = {"kernel": "lvt::k_hamming_batched<0,3,1,2> (masked 2-NN Hamming matcher, radius mode)", "bound": "hbm", "achieved": round(ach, 1),
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                  "avg_us": round(med, 2), "launch": {"B": B, "M": M, "N": N, "algorithmic_bytes": byts},
                  "traffic_source": traffic_src,
                  "device_copy_same_bytes_GBs": None if copy_gbs is None else round(copy_gbs, 1),
                  "frac_of_device_copy": None if copy_gbs is None else round(ach / copy_gbs, 4),
                  "note": "B independent KITTI-nominal problems per launch; algorithmic bytes 40(M+N)+N+16M each (SURVEY 8d); average of 5 "
                          "back-to-back launches between two HIP events on the launch stream, median of 7 such timings after 80 warm-up launches; traffic = "
                          "2*FETCH_SIZE + WRITE_SIZE of the same launch from the committed rocprofv3 PMC passes"}
        except Exception as e:  # noqa: BLE001
            hb = {"error": str(e)}
        # ---- CPU baseline: the oracle (port of the reference path), 2 threads like the reference, same frames
        cpu = None
        if not args.no_cpu and world_size == 1:  # reported at N = 1 only (rank 0's host cores)
            from oracle import pyoracle as O
            nf = min(n_frames, args.cpu_frames)
            host = frames[:nf, :, :, :W].contiguous().cpu().numpy()
            orc = O.Oracle(prm, 1, threads=2)
            tc = time.perf_counter()
            done = 0
            for i in range(nf):
                orc.track(host[i, 0], host[i, 1])
                done += 1
                if time.perf_counter() - tc > 30.0:
                    break
            tcpu = time.perf_counter() - tc
            cpu = {"value": round(done / tcpu, 2), "unit": "frames/s", "cores": 2, "kind": "port",
                   "sample": f"first {done} stereo pairs of the rank-0 sequence, oracle/liblvt_oracle.so with the reference's 2-thread "
                             f"left/right split; host has {os.cpu_count()} logical cores"}
            # 8 independent sequences side by side, 2 threads each (ctypes releases the GIL inside the oracle);
            # every instance tracks the same frames -- the aggregate rate is what 16 cores of this host deliver
            try:
                import threading
                n_par = min(8, max(1, (os.cpu_count() or 2) // 2))
                npf = min(nf, 120)
                orcs = [O.Oracle(prm, 1, threads=2) for _ in range(n_par)]
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
                cpu["parallel"] = {"sequences": n_par, "cores": 2 * n_par, "value": round(n_par * npf / tpar, 2), "unit": "frames/s",
                                   "sample": f"{n_par} oracle instances x the first {npf} stereo pairs, concurrently"}
            except Exception as e:  # noqa: BLE001
                cpu["parallel"] = {"error": str(e)}
        fps = world_size * K / elapsed
        result = {
            "metric": "stereo frames/sec (KITTI-shaped 1241x376), per-frame SE3 vs CPU ref",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world_size, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/f64", "data": "synthetic",
            "config": {"workload": "KITTI seq 00-shaped synthetic stereo sequence (1241x376, vo_config.yaml), one sequence per GPU, "
                                   "one lvt_track per step, frames resident in HBM",
                       "sequences_per_gpu": 1, "frames_in_flight": DEPTH, "ordering": vo.ordering(), "parallelism": f"{world_size} independent sequences, no collective"},
            "tracking": {"lost_frames": n_lost, "features_left": counts["n_left"], "map_size": counts["map_size"],
                         "matches": counts["n_matches"], "error": err},
            "roofline": hb, "roofline_pipeline_dominant": roofline_dom,
            "frame_algorithmic_bytes": None if frame_bytes is None else round(frame_bytes),
            "frame_hbm_frac": None if frame_bytes is None else round(frame_bytes * fps / world_size / 1e9 / HBM_PEAK_GBS, 6),
            "kernels": kernels, "cpu_baseline": cpu,
        }
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
