#!/bin/bash
# lock-step batch throughput by batch size and frames in flight: bash tools/batch_sweep.sh "<sizes>" "<depths>"
for d in ${2:-3}; do for n in ${1:-16}; do
  python bench.py --steps 200 --warmup 10 --seqs-per-gpu $n --depth $d --skip cpu,sync,lists_ab,kernels,batch,configs,pmc 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('seqs $n depth $d:', r['value'], r['ms_per_step'], r['tracking']['frames_not_tracking'], r['tracking_error_string'])"
done; done
