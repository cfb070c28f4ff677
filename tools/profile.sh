#!/bin/bash
# rocprofv3 kernel-trace summary of the benchmark (run on the GPU box through gpurun).
# Raw traces stay in /tmp; only the per-kernel stats tables are copied under gpurun_out/ (-> profiles/).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r01}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_single /tmp/prof_batch
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_single -o bench -- python $ROOT/bench.py --steps 300 --warmup 20 --no-cpu --profile-steps 0 --depth 2 > $OUT/bench_single.json 2> $OUT/bench_single.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_batch -o bench -- python $ROOT/bench.py --steps 100 --warmup 10 --seqs-per-gpu 16 --depth 2 > $OUT/bench_batch16.json 2> $OUT/bench_batch16.err
for d in single batch; do
  f=$(find /tmp/prof_$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$d.csv && echo "== $d: $f" && head -32 "$f"
done
tail -c 600 $OUT/bench_single.json; echo; tail -c 400 $OUT/bench_batch16.json; echo
find /tmp/prof_single | head -20
