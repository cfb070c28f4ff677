#!/bin/bash
# rocprofv3 per-kernel averages of a lock-step batch: bash tools/kstats_batch.sh [sequences] [frames in flight]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o bench -- python /root/repo/bench.py --steps 60 --warmup 10 --seqs-per-gpu ${1:-16} --depth ${2:-3} --skip cpu,sync,lists_ab,kernels,batch,configs,roofline > /tmp/pk.json 2>/dev/null
python - <<'PY'
import csv, glob, json
f = glob.glob("/tmp/pk/**/*kernel_stats.csv", recursive=True)[0]
print("fps under trace", json.loads(open("/tmp/pk.json").read().strip().splitlines()[-1])["value"])
for r in list(csv.reader(open(f)))[1:]:
    if "lvt::" in r[0]: print("  %-46s %5s %8.1f us" % (r[0].replace("void lvt::", "").replace("lvt::", "").split("(")[0][:46], r[1], float(r[3]) / 1e3))
PY
