#!/bin/bash
# SQ counters per kernel of a short lock-step batch run (counters only; event ordering): usage tools/pmc_kernels.sh "<counters>" [seqs] [steps]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CNT=${1:-SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES}
SEQS=${2:-16}; STEPS=${3:-10}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
LVT_AMD_ORDERING=events timeout 900 rocprofv3 --pmc $CNT --output-format csv -d /tmp/pk -o b -- python $ROOT/bench.py --steps $STEPS --warmup 3 --seqs-per-gpu $SEQS --depth 3 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pk/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f[0])):
    k = r.get("Kernel_Name", "")
    if "lvt::" not in k: continue
    a = agg[k.split("(")[0][:48]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
names = sorted({c for k in agg for c in agg[k]})
print("%-50s %6s " % ("kernel (per dispatch)", "n") + " ".join("%16s" % n[-16:] for n in names))
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", [0, 0])[1]):
    n = max(v[0] for v in agg[k].values())
    print("%-50s %6d " % (k, n) + " ".join("%16.0f" % (agg[k][c][1] / max(agg[k][c][0], 1)) for c in names))
PY
