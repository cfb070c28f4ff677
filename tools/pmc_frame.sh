#!/bin/bash
# HBM-side traffic of ONE tracked stereo pair, per kernel: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, counters only) over a short
# single-sequence bench.py run with event ordering (counter collection serialises dispatches).  Extra environment (e.g. LVT_AMD_BRIEF_FROM_IMAGE=1) is inherited.
#   usage: [ENV=..] tools/pmc_frame.sh > out.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_fetch /tmp/pf_write
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pf_$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
  LVT_AMD_ORDERING=events timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -o b -- python $ROOT/bench.py --steps 30 --warmup 5 --skip kernels,device_resident,roofline,sync,batch,lists_ab,configs,cpu > $d.log 2>&1 || { echo "pass $c failed:"; tail -5 $d.log; }
done
python3 - <<'PY'
import csv, glob, collections
tot = {}
for name, d in (("FETCH_SIZE", "/tmp/pf_fetch"), ("WRITE_SIZE", "/tmp/pf_write")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = r.get("Kernel_Name", "")
        if "lvt::" not in k or r.get("Counter_Name") != name: continue
        a = agg[k.split("(")[0][:60]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    frames = max(n for n, _ in agg.values())
    print("== %s (KB per dispatch; %d frames)" % (name, frames))
    t = 0.0
    for k, (n, v) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("%-62s %5d %10.1f" % (k, n, v / n)); t += v / frames
    print("per tracked pair: %.1f KB" % t); tot[name] = t
raw = (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / 1024
cor = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / 1024
print("raw counters      FETCH + WRITE per pair: %.2f MB" % raw)
print("corrected     2 * FETCH + WRITE per pair: %.2f MB (algorithmic 1.223 MB: x %.2f)   [FETCH_SIZE reports exactly 0.5000 of the bytes streamed at 4, 8 and 16 bytes per lane and" % (cor, cor / 1.223))
print("               in k_score's 64-byte row-segment pattern, WRITE_SIZE 1.0000: tools/pmc_calib.sh, profiles/r06_fetch_write_calibration.txt]")
import json
print(json.dumps({"frame_traffic": {"FETCH_SIZE_KB_per_pair_raw": round(tot["FETCH_SIZE"], 1), "WRITE_SIZE_KB_per_pair": round(tot["WRITE_SIZE"], 1),
      "corrected_bytes_per_pair": round((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024), "algorithmic_bytes_per_pair": 1223000, "ratio": round(cor / 1.223, 2)}}))
PY
