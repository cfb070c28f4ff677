#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun).  Raw traces stay in /tmp; compact per-kernel
# summaries of OUR kernels (lvt::*) go to gpurun_out/prof_<tag>/ and from there into profiles/.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r02}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_*
BENCH="python $ROOT/bench.py --steps 400 --warmup 20 --skip kernels,sync,batch,lists_ab,configs,cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_batch -o bench -- python $ROOT/bench.py --steps 100 --warmup 10 --seqs-per-gpu 16 --depth 3 > $OUT/bench_batch16_under_rocprof.json 2> /dev/null
# PMC: separate passes, counters only (no trace domains) -- FETCH_SIZE and WRITE_SIZE cannot share a pass
# (counter collection serialises the dispatches of all queues: the pipeline must not use its polling gates -> LVT_AMD_ORDERING=events;
#  the timeout only guards the box)
LVT_AMD_ORDERING=events timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -o bench -- python $ROOT/bench.py --steps 30 --warmup 5 --skip kernels,device_resident,sync,batch,lists_ab,configs,cpu > $OUT/bench_under_pmc.json 2>&1
LVT_AMD_ORDERING=events timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -o bench -- python $ROOT/bench.py --steps 30 --warmup 5 --skip kernels,device_resident,sync,batch,lists_ab,configs,cpu > /dev/null 2>&1
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
def stats(src, dst):
    f = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)
    if not f: return
    rows = list(csv.reader(open(f[0])))
    keep = [rows[0]] + [r for r in rows[1:] if "lvt::" in r[0]]
    csv.writer(open(dst, "w")).writerows(keep)
    print("==", dst); [print(",".join(r[:4])) for r in keep]
stats("/tmp/prof_trace", out + "/kernel_stats_single.csv")
# per-dispatch durations of the two reported matcher instances (bench.py warms the clocks up with launches of a THIRD instance, <0,5,1,2>, so each
# reported instance's --stats row holds 3 untimed + the 35 reported launches)
f = glob.glob("/tmp/prof_trace/**/*kernel_trace.csv", recursive=True)
if f:
    rd = csv.DictReader(open(f[0]))
    rows = list(rd)
    for tag, pat in (("", "k_hamming_batched<0, 3,"), ("_row_mode", "k_hamming_batched<1, 1, 1, 2>")):   # (the warm-up instance is <0, 5, ...>)
        d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if pat in r["Kernel_Name"].replace("<0,3,", "<0, 3,")]
        d.sort()
        w = csv.writer(open(out + "/hamming_dispatch_durations%s.csv" % tag, "w")); w.writerow(["dispatch", "duration_ns"])
        for i, (_, ns) in enumerate(d): w.writerow([i, ns])
        if len(d) >= 35:
            last = [ns for _, ns in d[-35:]]
            print("== matcher dispatches%s:" % tag, len(d), "mean of all %.1f us, of the last 35 (the timed ones) %.1f us" % (sum(ns for _, ns in d) / len(d) / 1e3, sum(last) / 35e3))
stats("/tmp/prof_batch", out + "/kernel_stats_batch16.csv")
def pmc(src, name, dst):
    f = glob.glob(src + "/**/*counter_collection.csv", recursive=True)
    if not f: print("no counter file for", name); return
    agg = collections.defaultdict(lambda: [0, 0.0])
    rd = csv.DictReader(open(f[0]))
    for r in rd:
        k = r.get("Kernel_Name", "")
        if "lvt::" not in k or r.get("Counter_Name") != name: continue
        a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"])
    w = csv.writer(open(dst, "w")); w.writerow(["Kernel_Name", "Dispatches", name + "_sum", name + "_per_dispatch"])
    print("==", dst)
    for k, (n, v) in sorted(agg.items(), key=lambda x: -x[1][1]):
        w.writerow([k, n, v, v / n]); print(k[:60], n, round(v / n, 2))
pmc("/tmp/prof_fetch", "FETCH_SIZE", out + "/pmc_fetch_size.csv")
pmc("/tmp/prof_write", "WRITE_SIZE", out + "/pmc_write_size.csv")
PY
tail -c 1500 $OUT/bench_under_rocprof.json
