#!/bin/bash
# headline throughput of one sequence by frames in flight, at the default run length and at the driver's (--steps 20 --warmup 5)
for d in ${1:-2 3 4 5 6}; do
  for kw in "400 20" "20 5"; do set -- $kw
    python bench.py --steps $1 --warmup $2 --depth $d --skip cpu,sync,lists_ab,kernels,batch,configs,roofline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('depth $d steps $1:', r['value'], r['ms_per_step'])"
  done
done
