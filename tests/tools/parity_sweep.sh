#!/bin/bash
# long differential runs of the HIP path against the oracle (tests/tools/gpu_parity.py) -- run on the GPU box, prints one SUMMARY per run
cd "$(dirname "$0")/../.."
F=${1:-300}
for s in 0 1 2 3 4 5 6 7 8 9 10; do timeout 600 python tests/tools/gpu_parity.py --kind kitti --seed $s --frames $F --verbose 0 2>&1 | grep SUMMARY | sed "s/^/seed $s /"; done
for s in 0 1 2; do timeout 600 python tests/tools/gpu_parity.py --kind euroc --seed $s --frames 200 --verbose 0 2>&1 | grep SUMMARY | sed "s/^/seed $s /"; done
for s in 0 1 2; do timeout 600 python tests/tools/gpu_parity.py --kind tum --seed $s --frames 200 --verbose 0 2>&1 | grep SUMMARY | sed "s/^/seed $s /"; done
timeout 600 python tests/tools/gpu_parity.py --kind kitti --seed 3 --frames 120 --jump 60 --verbose 0 2>&1 | grep SUMMARY | sed "s/^/jump /"
