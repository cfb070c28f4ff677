#!/usr/bin/env python3
"""Sweep of the batched Hamming matcher over the micro-benchmark shapes of SURVEY 8(d): batch size, mask kind, descriptor
statistics.  Prints a markdown table (committed as profiles/*_hamming_sweep.md).
    python tools/hamming_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
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
    reps = 1 if mask == "none" else 5
    us = sorted(lvt_amd.hamming_match_batched(qd, qxy, td, txy, tf, r2, mode, H, W, out, launches=reps) for _ in range(4))[1]
    byts = B * (40.0 * (M + N) + N + 16.0 * M)
    cand = float((out[:, :, 0] >= 0).float().mean().item())
    return us, byts / us / 1e3, cand


def main():
    rows = []
    for B in (1, 64, 1024, 2048, 8192):
        rows.append((B, 1000, 1500, "radius25", "iid"))
    for B in (64, 2048):
        rows.append((B, 1000, 1500, "row+-2", "iid"))
    rows += [(2048, 1000, 1500, "radius25", "planted"), (2048, 1000, 1500, "radius25", "ties"), (2048, 256, 600, "radius25", "iid"),
             (2048, 1024, 1000, "radius25", "iid"), (2048, 1500, 1500, "radius25", "iid"), (64, 1000, 1500, "none", "iid")]
    print("| B | M | N | mask | descriptors | us / launch | algorithmic GB/s | % of 8 TB/s | queries with a match |")
    print("|---|---|---|---|---|---|---|---|---|")
    for B, M, N, mask, variant in rows:
        us, gbs, cand = run(B, M, N, mask, variant)
        print("| %d | %d | %d | %s | %s | %.1f | %.0f | %.1f | %.2f |" % (B, M, N, mask, variant, us, gbs, gbs / 80.0, cand))


if __name__ == "__main__":
    main()
