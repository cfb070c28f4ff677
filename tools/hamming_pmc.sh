#!/bin/bash
# VALU / LDS counters of the batched matcher (rocprofv3 --pmc, counters only, one small group per pass).
#   tools/hamming_pmc.sh [B] [out] [root of the tree whose tools/hamming_bench.py is profiled]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=${1:-2048}
OUT=${2:-$ROOT/gpurun_out/hamming_pmc.txt}
TREE=${3:-$ROOT}
cd /tmp && export TMPDIR=/tmp
: > $OUT
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM"; do
    rm -rf /tmp/hpmc
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/hpmc -o h -- python $TREE/tools/hamming_bench.py $B 1000 1500 0 > /dev/null 2>&1
    python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/hpmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
if f:
    for r in csv.DictReader(open(f[0])):
        if "k_hamming" in r.get("Kernel_Name", ""):
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open(sys.argv[1], "a") as o:
    for k, (n, v) in sorted(agg.items()):
        o.write("%s dispatches %d mean %.1f\n" % (k, n, v / n))
PY
done
cat $OUT
