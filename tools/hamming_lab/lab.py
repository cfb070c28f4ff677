#!/usr/bin/env python3
"""Development harness: time ONE instantiation of the batched matcher built from a kernel file and diff its output against the
shipped library.   usage: python tools/hamming_lab/lab.py <kernel.hip> [B] [--instance 'k_hamming_batched<0,3,1,2>'] [--has-b]
Builds tools/hamming_lab/_lab_<tag>.so when run on a box without it (hipcc cross-compiles), runs when a GPU is present."""
import argparse, ctypes as C, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
ap = argparse.ArgumentParser()
ap.add_argument("kernel")
ap.add_argument("B", nargs="?", type=int, default=8192)
ap.add_argument("--instance", default="k_hamming_batched<0, 3, 1, 2>")
ap.add_argument("--has-b", action="store_true")
ap.add_argument("--band", action="store_true", help="the kernel takes BandArgs (k_hamming_band.hip)")
ap.add_argument("--split", type=int, default=0, help="v6_split.hip: sub-problems per problem")
ap.add_argument("--threads", type=int, default=512, help="v6_split.hip: threads per workgroup")
ap.add_argument("--wpe", type=int, default=0, help="v6_split.hip: waves per SIMD the register allocation is held to (default: by LDS residency)")
ap.add_argument("--mode", type=int, default=0, help="0 = radius mode (r = 25), 1 = row mode")
ap.add_argument("--define", default="", help="extra -D for the build, e.g. LVT_ROW_WALK=2 (becomes part of the library's name)")
ap.add_argument("--build-only", action="store_true")
ap.add_argument("--ref", default=os.path.join(ROOT, "lvt_amd", "lib", "liblvt_c.so"), help="library whose matcher output is the reference (default: the shipped one)")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--stop", type=int, default=0)
ap.add_argument("--timeline", action="store_true", help="per-workgroup (start, end, CU) records: slot occupancy and gaps")
ap.add_argument("--pmc", action="store_true", help="one launch only (under rocprofv3 --pmc)")
ap.add_argument("--coherent", type=int, default=0, help="experiment: 1 = every query of a problem at the same position (all lanes read the same LDS addresses: no bank conflicts, "
                "perfectly balanced waves), 2 = queries sorted along x (neighbouring lanes read neighbouring list entries)")
a = ap.parse_args()
kern = os.path.abspath(a.kernel)
tag = os.path.splitext(os.path.basename(kern))[0]
if a.mode == 1 and a.instance == "k_hamming_batched<0, 3, 1, 2>":
    a.instance = "k_hamming_batched<1, 1, 1, 2>"
if a.split and not a.wpe:
    a.wpe = 6 if (a.mode == 0 and a.split == 2 and a.threads == 512) else 8   # (three 512-thread workgroups per CU: six waves per SIMD, 80 VGPRs)
so = os.path.join(HERE, "_lab_%s%s%s.so" % (tag, "_stop%d" % a.stop if a.stop else "", ("_m%d_s%d_t%d_w%d" % (a.mode, a.split, a.threads, a.wpe) if (a.split or a.mode) else "") + ("_" + a.define.replace("=", "") if a.define else "")))
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(kern):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-inline-asm", "-Wno-unused-value", "-shared",
           "-I" + os.path.join(ROOT, "lvt_amd", "csrc"), '-DLAB_KERNEL="%s"' % kern, "-DLAB_INSTANCE=" + a.instance.replace(" ", "")] + \
          (["-DLAB_HAS_B"] if a.has_b else []) + (["-DLAB_BAND"] if a.band else []) + ["-DLAB_MODE=%d" % a.mode] + (["-D" + a.define] if a.define else []) + \
          (["-DLAB_SPLIT=%d" % a.split, "-DLAB_THREADS=%d" % a.threads, "-DLAB_WPE=%d" % a.wpe] if a.split else []) + ["-DLAB_STOP=%d" % a.stop] + ["-o", so, os.path.join(HERE, "lab.hip")]
    subprocess.check_call(cmd)
if a.build_only:
    sys.exit(0)
import torch
B, M, N, W, H = a.B, 1000, 1500, 1241, 376
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1234)
qd = torch.randint(0, 256, (B, M, 32), dtype=torch.uint8, device=dev, generator=g)
td = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=dev, generator=g)
qxy = (torch.rand((B, M, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev)).contiguous()
txy = torch.floor(torch.rand((B, N, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev)).contiguous()
if a.coherent == 1:
    qxy = qxy[:, :1, :].expand(B, M, 2).contiguous()
elif a.coherent == 2:
    qxy = torch.gather(qxy, 1, torch.argsort(qxy[:, :, 0], dim=1).unsqueeze(-1).expand(B, M, 2)).contiguous()
tf = (torch.rand((B, N), device=dev, generator=g) < 0.02).to(torch.uint8)   # a few flagged train features
out = torch.zeros((B, M, 4), dtype=torch.int32, device=dev)
ref = torch.zeros_like(out)
p = lambda t: C.c_void_p(t.data_ptr())
L = C.CDLL(so)
L.lab_run.restype = C.c_float
R = C.CDLL(a.ref)
R.lvt_amd_hamming_match_batched_n.restype = C.c_float
R.lvt_amd_hamming_match_batched_n(p(qd), p(qxy), p(td), p(txy), p(tf), B, M, N, C.c_float(625.0), a.mode, H, W, p(ref), None, 1)
def run(n, dbg=None):
    return L.lab_run(p(qd), p(qxy), p(td), p(txy), p(tf), B, M, N, C.c_float(625.0), H, W, p(out), n, dbg)
info = (C.c_longlong * 16)(); info[14] = 777
run(1, info)
torch.cuda.synchronize()
same = bool(torch.equal(out, ref))
print("%s: output == shipped library: %s" % (tag, same))
if not same:
    bad = (out != ref).any(dim=2).nonzero()
    print("  differing queries:", bad.shape[0], "first:", bad[:4].tolist(), out[bad[0, 0], bad[0, 1]].tolist(), ref[bad[0, 0], bad[0, 1]].tolist())
if a.pmc:
    run(3); torch.cuda.synchronize(); sys.exit(0)
for _ in range(8):
    run(10)
for rep in range(a.reps):
    us = sorted(run(5) for _ in range(7))
    byts = B * (40.0 * (M + N) + N + 16.0 * M)
    print("%s B=%d: median %.1f us  min %.1f  -> %.3f of 8 TB/s" % (tag, B, us[3], us[0], byts / us[3] / 1e3 / 8000.0))
dbg = (C.c_longlong * 16)()
run(1, dbg)
d = list(dbg)
if d[0]:
    if d[7] == 0: d[7] = d[5]
    print("  phases: load %d count %d scan %d scatter+qcount %d qsort %d stageA+sort2 %d stageB %d total %d" %
          (d[1] - d[0], d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4], d[7] - d[5], d[6] - d[7], d[6] - d[0]))
    if d[9] > d[8]:
        print("  workgroup life %.2f us (100-MHz clock) -> %.0f MHz shader clock" % ((d[9] - d[8]) / 100.0, (d[6] - d[0]) / ((d[9] - d[8]) / 100.0)))

if a.timeline:
    import numpy as np
    tl = (C.c_longlong * (16 + 4 * B))()
    tl[15] = 12345
    run(1, tl)
    arr = np.frombuffer(tl, dtype=np.int64)[16:].reshape(B, 4).copy()
    arr = arr[arr[:, 0] != 0]   # (a persistent kernel writes one record per resident workgroup)
    import collections
    hwc = (arr[:, 3] >> 16) * 4096 + ((arr[:, 3] >> 8) & 0xF) + (((arr[:, 3] >> 12) & 0x1) << 4) + (((arr[:, 3] >> 13) & 0x7) << 5)
    _ids = np.nonzero(np.frombuffer(tl, dtype=np.int64)[16:].reshape(B, 4)[:, 0])[0]
    _bycu = collections.defaultdict(list)
    for _i, _c in zip(_ids.tolist(), hwc.tolist()): _bycu[_c].append(_i)
    print('  block ids sharing a CU (first 12 CUs):', [v for _, v in sorted(_bycu.items())][:12])
    if len(_ids) == 512:
        _life = (arr[:, 1] - arr[:, 0]) / 100.0
        print('  life of the resident workgroups: first 256 (older) mean %.1f us, last 256 (younger) mean %.1f us' % (_life[_ids < 256].mean(), _life[_ids >= 256].mean()))
    print('  records', len(arr), 'workgroups per CU histogram:', sorted(collections.Counter(collections.Counter(hwc.tolist()).values()).items()))
    t0 = arr[:, 0].min()
    st, e0, e1, hw = (arr[:, 0] - t0) / 100.0, (arr[:, 1] - t0) / 100.0, (arr[:, 2] - t0) / 100.0, arr[:, 3]
    end = np.maximum(e0, e1)
    cu = (hw >> 16) * 4096 + ((hw >> 8) & 0xF) + (((hw >> 12) & 0x1) << 4) + (((hw >> 13) & 0x7) << 5)
    print("  timeline: kernel span %.1f us; workgroup life mean %.2f us (wave 0) / %.2f (to the end of wave 15); distinct CUs %d" %
          (end.max(), (e0 - st).mean(), (end - st).mean(), len(set(cu.tolist()))))
    gaps, conc = [], []
    for c in set(cu.tolist()):
        idx = np.where(cu == c)[0]
        o = idx[np.argsort(st[idx])]
        # two slots per CU: assign greedily
        slots = [None, None]
        for i in o:
            k = 0 if (slots[0] is None or end[slots[0]] <= st[i] + 1e-9) else 1
            if slots[k] is not None:
                gaps.append(st[i] - end[slots[k]])
            slots[k] = i
    gaps = np.array(gaps if gaps else [0.0])
    print("  gap between a workgroup's end and its successor's start on the same CU slot: mean %.2f us  p10 %.2f  p50 %.2f  p90 %.2f  (n=%d)" %
          (gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 50), np.percentile(gaps, 90), len(gaps)))
