#!/bin/bash
# registers / spills / scratch of every k_hamming_batched template instance in the product library (code-object metadata of a device-only compile)
#   usage: tools/hamming_instances.sh > profiles/r05_hamming_instances_registers.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fvisibility=hidden -Wno-unused-function -Wno-inline-asm --cuda-device-only -S -o $T/host.s $ROOT/lvt_amd/csrc/lvt_host.hip 2>/dev/null
python3 - $T/host.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
rows = []
for blk in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    m = re.match(r"_ZN3lvt17k_hamming_batchedILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEE", name)
    if not m: continue
    g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
    rows.append((tuple(int(x) for x in m.groups()), g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
print("# k_hamming_batched<MODE, NSP, QPT, TPT>: MODE 0 radius / 1 row; NSP ranges kept in registers (3: csr 1, 5: csr 2, 0: any); QPT = ceil(M / 1024); TPT = ceil(N / 1024)")
for (k, v, sp, ss, sc) in sorted(rows):
    print("k_hamming_batched<%d,%d,%d,%d>  vgprs %3d  spilled %3d  sgpr-spilled %3d  scratch %4d B" % (*k, v, sp, ss, sc))
PY
rm -rf $T
