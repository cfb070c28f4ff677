mkdir -p gpurun_out/c9
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c9/pytest_gpu.txt 2>&1
tail -5 gpurun_out/c9/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/c9/bench_driver_args.json 2> gpurun_out/c9/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c9/bench_driver_args.json").read().strip().splitlines()[-1])
print("value", d["value"], "dev", d["device_resident"]["second"]["fps"], "cfg", json.dumps(d.get("configs")), "err", {k: v for k, v in d.items() if k.endswith("_error")})
print("roofline", d["roofline"]["frac"], d["roofline"]["row_mode"]["frac"], "sync p50", d["sync"]["lvt_track_host_ms"]["p50"])
print("batch_sweep", [(r["seqs"], r["fps"]) for r in d["batch_sweep"]])
print("kernels", [(k["kernel"][:24], k["avg_us"]) for k in d["kernels"]])
PY
OUT=gpurun_out/c9/scale STEPS=20 WARMUP=5 timeout 900 bash tools/scale.sh --share-gpu > gpurun_out/c9/scale_share_gpu.txt 2>&1
tail -12 gpurun_out/c9/scale_share_gpu.txt
