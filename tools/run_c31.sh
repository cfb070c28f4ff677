for m in 0 1 0 1; do timeout 300 python tools/hamming_lab/lab.py lvt_amd/csrc/k_hamming.hip --mode $m 2>&1 | grep "B=8192\|==" ; done
