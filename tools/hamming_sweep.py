#!/usr/bin/env python3
"""Sweep of the batched Hamming matcher over the micro-benchmark shapes of SURVEY 8(d): B in {1, 64, 1024, 8192} x (M, N) in
{(256, 600), (1024, 1000), (1500, 1500)} (+ the KITTI-nominal (1000, 1500)) x mask {radius 25 px, row +-2, none} x descriptors
{iid, planted, ties}.  Prints a markdown table (committed as profiles/r0N_hamming_sweep.md) with the template instance every row runs
and whether that instance spills registers (profiles/r0N_hamming_instances_registers.txt, read when present).
    python tools/hamming_sweep.py [--quick]"""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import torch

import lvt_amd

W, H = 1241, 376
dev = torch.device("cuda:0")


def make(B, M, N, variant, g):
    td = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=dev, generator=g)
    qd = torch.randint(0, 256, (B, M, 32), dtype=torch.uint8, device=dev, generator=g)
    txy = torch.floor(torch.rand((B, N, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev)).contiguous()
    qxy = (torch.rand((B, M, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev)).contiguous()
    if variant == "planted":  # query = a train row with 10 flipped bits, placed within 3 px of it
        src = torch.randint(0, N, (B, M), device=dev, generator=g)
        qd = torch.gather(td, 1, src[:, :, None].expand(B, M, 32)).clone()
        qd[:, :, 0] ^= 0x1F
        qd[:, :, 7] ^= 0x1F
        qxy = (torch.gather(txy, 1, src[:, :, None].expand(B, M, 2)) + (torch.rand((B, M, 2), device=dev, generator=g) * 6 - 3)).contiguous()
    if variant == "ties":  # 16 prototypes: massive distance ties
        proto = torch.randint(0, 256, (16, 32), dtype=torch.uint8, device=dev, generator=g)
        td = proto[torch.randint(0, 16, (B, N), device=dev, generator=g)]
        qd = proto[torch.randint(0, 16, (B, M), device=dev, generator=g)]
    tf = torch.zeros((B, N), dtype=torch.uint8, device=dev)
    out = torch.zeros((B, M, 4), dtype=torch.int32, device=dev)
    return qd.contiguous(), qxy, td.contiguous(), txy, tf, out


def instance(mask, M, N):
    """the template instance lvt_amd_hamming_match_batched picks (lvt_host.hip): <MODE, NSP, QPT, TPT>"""
    qpt, tpt = -(-M // 1024), -(-N // 1024)
    mode, nsp = (1, 1) if mask == "row+-2" else (0, 3) if mask == "radius25" else (0, 0)
    return "<%d,%d,%d,%d>" % (mode, nsp, qpt, tpt)


def spills():
    """{instance: spilled VGPRs} from the newest profiles/r*_hamming_instances_registers.txt"""
    pdir = os.path.join(HERE, "..", "profiles")
    files = sorted(f for f in os.listdir(pdir) if f.endswith("_hamming_instances_registers.txt"))
    out = {}
    if files:
        for line in open(os.path.join(pdir, files[-1])):
            m = re.match(r"k_hamming_batched(<[0-9,]+>)\s+vgprs\s+(\d+)\s+spilled\s+(\d+)", line)
            if m:
                out[m.group(1)] = int(m.group(3))
    return out


def run(B, M, N, mask, variant):
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    qd, qxy, td, txy, tf, out = make(B, M, N, variant, g)
    if mask == "radius25":
        mode, r2 = 0, 625.0
    elif mask == "row+-2":
        mode, r2 = 1, 0.0
    else:  # no mask: a radius that covers the image (every query sees every train feature)
        mode, r2 = 0, float((W + H) ** 2)
    reps = 1 if mask == "none" else (5 if B >= 1024 else 20)
    for _ in range(2 if mask == "none" else 4):
        lvt_amd.hamming_match_batched(qd, qxy, td, txy, tf, r2, mode, H, W, out, launches=reps)
    us = sorted(lvt_amd.hamming_match_batched(qd, qxy, td, txy, tf, r2, mode, H, W, out, launches=reps) for _ in range(5))[2]
    byts = B * (40.0 * (M + N) + N + 16.0 * M)
    cand = float((out[:, :, 0] >= 0).float().mean().item())
    return us, byts / us / 1e3, cand


def main():
    quick = "--quick" in sys.argv
    shapes = [(256, 600), (1024, 1000), (1500, 1500), (1000, 1500)]
    rows = []
    for B in (1, 64, 1024, 8192):
        for M, N in shapes:
            for mask in ("radius25", "row+-2", "none"):
                if mask == "none" and B > 64:
                    continue   # (2.25 M descriptor distances per problem: the no-mask rows stop at 64 problems)
                for variant in ("iid", "planted", "ties"):
                    if quick and (variant != "iid" or (M, N) != (1000, 1500)):
                        continue
                    rows.append((B, M, N, mask, variant))
    sp = spills()
    print("| B | M | N | mask | descriptors | instance | spilled VGPRs | us / launch | algorithmic GB/s | % of 8 TB/s | queries with a match |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for B, M, N, mask, variant in rows:
        us, gbs, cand = run(B, M, N, mask, variant)
        inst = instance(mask, M, N)
        print("| %d | %d | %d | %s | %s | `%s` | %s | %.1f | %.0f | %.1f | %.2f |" % (B, M, N, mask, variant, inst, sp.get(inst, "?"), us, gbs, gbs / 80.0, cand), flush=True)


if __name__ == "__main__":
    main()
