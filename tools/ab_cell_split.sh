# A/B of the co-operating cell split (LVT_AMD_CELL_SPLIT = 0 | 2) on one box: parity subset, k_cells phases, headline
SK="--skip kernels,roofline,pmc,sync,batch,lists_ab,configs,cpu --no-cpu"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('device_resident') or {}; print('$1', d['value'], (r.get('first') or {}).get('fps'), (r.get('second') or {}).get('fps'))"; }
export LVT_AMD_CELL_SPLIT=2
echo "== split 2 parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "kitti_half or kitti_full or kitti_dense_anms or kitti_low_corner or kitti_jump or euroc or kitti_hard" 2>&1 | tail -3
for s in 0 2 0 2 0 2; do
  export LVT_AMD_CELL_SPLIT=$s
  echo "== split $s"; python tools/cells_phases.py kitti 4 2>/dev/null | grep "k_cells" | tail -1 | cut -c1-420
  python bench.py --steps 400 --warmup 40 $SK 2>/dev/null | show "steady split=$s"
  python bench.py --steps 20 --warmup 5 $SK 2>/dev/null | show "driver-args split=$s"
done
python tools/sync_latency.py 2>/dev/null | tail -3
LVT_AMD_CELL_SPLIT=0 python tools/sync_latency.py 2>/dev/null | tail -3
