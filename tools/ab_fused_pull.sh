SK="--skip kernels,roofline,pmc,sync,batch,lists_ab,configs,cpu --no-cpu"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], (d.get('device_resident') or {}).get('fps'))"; }
for f in 1 0; do LVT_AMD_HOST_TIMING=1 LVT_AMD_FUSED_PULL=$f python bench.py --steps 400 --warmup 40 $SK 2>&1 | grep -i "host\|enq" | head -5; done
for f in 1 0; do LVT_AMD_FUSED_PULL=$f python bench.py --steps 400 --warmup 40 --skip kernels,roofline,pmc,sync,batch,lists_ab,configs,cpu 2>/dev/null | show "fused=$f steady+devres"; done
