#!/bin/bash
# Regenerates the round's measurement artefacts (R=r03) on the GPU box; results under gpurun_out/refresh/ (copy into profiles/).
# Each rocprofv3 step runs python3 directly (no wrapper between the profiler and the program).
set -e
R=${R:-r03}
OUT=/root/repo/gpurun_out/refresh
mkdir -p $OUT
cd /root/repo
echo "[1] bench (no profiler)"; timeout -k 10 600 python3 bench.py > $OUT/${R}_bench_n1.json 2> $OUT/bench.err
echo "[2] kernel trace of the driver's command"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o ${R} -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_n1_under_rocprof.json 2> $OUT/kt.err
cd /root/repo
python3 tools/prof_summary.py /tmp/kt 60 > $OUT/${R}_bench_n1_kernel_summary.txt
cp /tmp/kt/${R}_kernel_stats.csv $OUT/${R}_bench_n1_kernel_stats.csv
echo "[3] PMC pass (FETCH_SIZE only)"
cd /tmp && rm -rf /tmp/pmc
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc -o bench -- python3 /root/repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline --ttft-prompts 1 --no-speculation > $OUT/pmc_bench.json 2> $OUT/pmc.err
cd /root/repo
python3 tools/pmc_summary.py /tmp/pmc $OUT/${R}_gemv_traffic.json > /dev/null
echo "[4] Qwen2.5-7B INT8"; timeout -k 10 600 python3 bench.py --model qwen25_7b --no-cpu-baseline > $OUT/${R}_bench_n1_qwen25_7b_int8.json 2> $OUT/qwen.err
for T in 2 4 8; do
  echo "[5] TP $T loopback kernel trace"
  cd /tmp && rm -rf /tmp/tp$T
  timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tp$T -o tp -- python3 /root/repo/bench.py --gpus $T --tp-loopback --steps 8 --warmup 2 --ttft-prompts 1 --no-cpu-baseline > $OUT/tp${T}_bench.json 2> $OUT/tp$T.err
  cd /root/repo
  python3 tools/tp_kernel_summary.py /tmp/tp$T $T > $OUT/${R}_tp${T}_loopback_kernels.json
done
echo "[6] 70B TP 8 loopback kernel trace"
cd /tmp && rm -rf /tmp/tp70
timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tp70 -o tp -- python3 /root/repo/bench.py --model llama33_70b --gpus 8 --tp-loopback --steps 8 --warmup 2 --ttft-prompts 1 --no-cpu-baseline > $OUT/tp70_bench.json 2> $OUT/tp70.err
cd /root/repo
python3 tools/tp_kernel_summary.py /tmp/tp70 8 > $OUT/${R}_tp8_llama33_70b_loopback_kernels.json
ls -la $OUT
echo done
