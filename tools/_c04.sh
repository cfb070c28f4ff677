mkdir -p gpurun_out/c04
timeout 200 python tools/lists_phases.py 16 30 > gpurun_out/c04/phases.txt 2>&1; tail -2 gpurun_out/c04/phases.txt
timeout 300 bash tools/kstats_batch.sh 16 3 > gpurun_out/c04/kstats16.txt 2>&1; cat gpurun_out/c04/kstats16.txt
timeout 300 bash tools/batch_sweep.sh "1 4 16 32 64" 3 > gpurun_out/c04/sweep.txt 2>&1; cat gpurun_out/c04/sweep.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c04/tests.txt 2>&1
tail -5 gpurun_out/c04/tests.txt
