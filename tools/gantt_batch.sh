#!/bin/bash
# every dispatch of two steady-state steps of a lock-step batch, in start order, from a rocprofv3 kernel trace: bash tools/gantt_batch.sh [sequences] [depth]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg
rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o bench -- python /root/repo/bench.py --steps 60 --warmup 10 --seqs-per-gpu ${1:-16} --depth ${2:-2} --skip cpu,sync,lists_ab,kernels,batch,configs > /dev/null 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/pg/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "lvt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pnp = [i for i, r in enumerate(rows) if "k_pnp" in r["Kernel_Name"]]
a, b = pnp[len(pnp) // 2], pnp[len(pnp) // 2 + int(__import__("os").environ.get("STEPS", "2"))]
t0 = int(rows[a]["Start_Timestamp"])
def short(n): return re.sub(r"\(.*", "", n.replace("void lvt::", "").replace("lvt::", ""))[:34]
qs = sorted({r["Queue_Id"] for r in rows})
for r in rows[a:b + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    col = qs.index(r["Queue_Id"])
    print("%8.1f %8.1f  %6.1f  q%d %s%s  grid %s" % (s, e, e - s, col, "    " * col, short(r["Kernel_Name"]), r.get("Grid_Size", "")))
PY
