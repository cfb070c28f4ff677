mkdir -p gpurun_out/c03
timeout 900 python -m pytest tests -m gpu -x -q -k "binned or lockstep or row_lists or pooled or batch" > gpurun_out/c03/tests.txt 2>&1
tail -5 gpurun_out/c03/tests.txt
timeout 200 python tools/lists_phases.py 16 30 > gpurun_out/c03/phases.txt 2>&1; tail -2 gpurun_out/c03/phases.txt
timeout 300 bash tools/kstats_batch.sh 16 3 > gpurun_out/c03/kstats16.txt 2>&1; cat gpurun_out/c03/kstats16.txt
timeout 300 bash tools/batch_sweep.sh "16 64" 3 > gpurun_out/c03/sweep.txt 2>&1; cat gpurun_out/c03/sweep.txt
