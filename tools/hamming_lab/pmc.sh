#!/bin/bash
# usage: pmc.sh <kernel.hip> <stop> "<counters>" [B] [extra lab.py arguments]  -> prints mean counter values per dispatch
HERE=$(cd "$(dirname "$0")" && pwd)
K=$1; STOP=$2; CNT=$3; B=${4:-4096}; shift 3; [ $# -gt 0 ] && shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/labpmc
timeout 300 rocprofv3 --pmc $CNT --output-format csv -d /tmp/labpmc -o h -- python $HERE/lab.py $HERE/$K $B --stop $STOP --pmc "$@" > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/labpmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    if "k_hamming" in r.get("Kernel_Name", ""):
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("  ".join("%s %.0f" % (k, v / n) for k, (n, v) in sorted(agg.items())))
PY
