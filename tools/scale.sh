#!/bin/bash
# The 8-GPU measurement, one command away (SURVEY 8e / BASELINE.json cfg 5: 8 independent sequences, one process per GPU, no collective).
#   tools/scale.sh                      on an N-GPU node: --gpus 1 2 4 8 (as many as the node has), one sequence per GPU           (weak scaling)
#                                       and --gpus 1 2 4 8 --total-seqs 8: the SAME eight sequences on 1 / 2 / 4 / 8 GPUs           (cfg 5)
#   tools/scale.sh --share-gpu          the same launches on a box with FEWER GPUs: ranks share them -- a rehearsal of the launch, rendezvous, timing
#                                       and reduction path, NOT a scaling measurement (the line says so in config.workload)
#   STEPS / WARMUP (default 20 / 5) are the driver's; the N = 1 line is byte for byte the driver's own command, so SCALE's N = 1 must agree with BENCH.
# Every line printed is bench.py's ONE JSON line; the summary underneath lists value, per-rank rates and frames not TRACKING.  No efficiency is computed
# here: the driver computes it from the values.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
SHARE=""
[ "$1" = "--share-gpu" ] && SHARE="--share-gpu"
STEPS=${STEPS:-20}; WARMUP=${WARMUP:-5}
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
OUT=${OUT:-gpurun_out/scale}; mkdir -p "$OUT"
run() {  # tag, gpus, extra args
    local tag=$1 n=$2; shift 2
    if [ -z "$SHARE" ] && [ "$n" -gt "$NGPU" ]; then echo "# $tag: needs $n GPUs, this node has $NGPU -- skipped (use --share-gpu for a rehearsal)"; return; fi
    local extra="$* $SHARE"
    [ "$n" = 1 ] && [ -z "$*" ] && extra=""          # N = 1: exactly the driver's command
    local cmd="python3 bench.py --gpus $n --steps $STEPS --warmup $WARMUP $extra"
    [ "$n" -gt 1 ] && cmd="$cmd --skip sync,batch,lists_ab,configs,cpu,kernels,roofline"
    echo "# $tag: $cmd"
    $cmd > "$OUT/$tag.json" 2> "$OUT/$tag.err" || echo "# $tag: rc $?"
    tail -n 1 "$OUT/$tag.json"
}
for n in 1 2 4 8; do run "n${n}_one_sequence_per_gpu" $n; done
for n in 1 2 4 8; do run "n${n}_total_seqs_8" $n --total-seqs 8; done
python3 - "$OUT" <<'PY'
import glob, json, os, sys
print("\n# summary (value = all ranks' frames / MAX of the rank-local durations)")
for f in sorted(glob.glob(os.path.join(sys.argv[1], "n*.json")), key=lambda p: (("total" in p), int(os.path.basename(p)[1:].split("_")[0]))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(os.path.basename(f), "no JSON line:", e); continue
    print("%-32s n_gpus %d seqs/gpu %d value %9.1f frames/s  per-rank %s  not-tracking %d%s" % (
        os.path.basename(f)[:-5], d["n_gpus"], d["config"]["sequences_per_gpu"], d["value"], d["timing"]["per_rank_fps"], d["tracking"]["frames_not_tracking"],
        "  [REHEARSAL: ranks share GPUs]" if "--share-gpu" in d["config"]["workload"] else ""))
PY
