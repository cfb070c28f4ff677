#!/usr/bin/env python3
"""usage: python tools/kstats.py <rocprofv3 output dir> [out.csv]
compact per-kernel table (name, calls, total us, average us) of OUR kernels from a `rocprofv3 --kernel-trace --stats` run"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "lvt::" in r["Name"]]
out = csv.writer(open(sys.argv[2], "w")) if len(sys.argv) > 2 else None
if out: out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs"])
for r in rows:
    if out: out.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"]])
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    print("%-50s %5s  total %9.1f us  avg %8.1f us" % (name[:50], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
