#!/bin/bash
set -e
OUT=/root/repo/gpurun_out/refresh
mkdir -p $OUT
cd /root/repo
timeout -k 10 600 python3 bench.py > $OUT/r03_bench_n1.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o r03 -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r03_bench_n1_under_rocprof.json 2> $OUT/kt.err
cd /root/repo
python3 tools/prof_summary.py /tmp/kt 60 > $OUT/r03_bench_n1_kernel_summary.txt
cp /tmp/kt/r03_kernel_stats.csv $OUT/r03_bench_n1_kernel_stats.csv
tail -c 600 $OUT/r03_bench_n1.json
