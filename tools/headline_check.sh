# quick headline check: host frames (pinned) and device-resident, steady state and at the driver's arguments
SK="--skip kernels,roofline,pmc,sync,batch,lists_ab,configs,cpu --no-cpu"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('device_resident') or {}; print('$1', d['value'], (r.get('first') or {}).get('fps'), (r.get('second') or {}).get('fps'))"; }
for i in 1 2 3; do python bench.py --steps 400 --warmup 40 $SK 2>/dev/null | show "steady"; done
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 $SK 2>/dev/null | show "driver-args"; done
