#!/bin/bash
# what rocprofv3's FETCH_SIZE / WRITE_SIZE report for 1 GiB streamed with 4 / 8 / 16 bytes per lane (tools/lab/fetch_calib.hip): reported bytes / true bytes per kernel
#   usage: tools/pmc_calib.sh > profiles/rNN_fetch_write_calibration.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -x $ROOT/tools/lab/bin/fetch_calib ] || { mkdir -p $ROOT/tools/lab/bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $ROOT/tools/lab/bin/fetch_calib $ROOT/tools/lab/fetch_calib.hip; }
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc_fetch /tmp/pc_write
timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pc_fetch -o c -- $ROOT/tools/lab/bin/fetch_calib > /tmp/pc_fetch.log 2>&1 || tail -5 /tmp/pc_fetch.log
timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pc_write -o c -- $ROOT/tools/lab/bin/fetch_calib > /tmp/pc_write.log 2>&1 || tail -5 /tmp/pc_write.log
python3 - <<'PY'
import csv, glob, collections
GIB = float(1 << 30)
for name, d in (("FETCH_SIZE", "/tmp/pc_fetch"), ("WRITE_SIZE", "/tmp/pc_write")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") == name: agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024.0)
    print("== %s: reported bytes / 1 GiB actually streamed (last of 3 launches; kernels that do not touch that direction should read ~0)" % name)
    for k, v in sorted(agg.items()):
        print("  %-40s %.4f" % (k[:40], v[-1] / GIB))
PY
