#!/bin/bash
# dev: rocprofv3 kernel trace of tools/prefill_once.py per bucket and mode -> gpurun_out/prefill_<bucket>_a<mode>.txt
#   bash tools/prof_prefill.sh "256 512 2048" "0 1"
cd /tmp && export TMPDIR=/tmp
for b in ${1:-256 512 1024 2048}; do for a in ${2:-0 1}; do
  rm -rf /tmp/ppf
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ppf -o pf -- python3 /root/repo/tools/prefill_once.py $b $a 32 3 > /root/repo/gpurun_out/prefill_${b}_a${a}.log 2>&1 || exit 1
  ( tail -1 /root/repo/gpurun_out/prefill_${b}_a${a}.log; python3 /root/repo/tools/prof_summary.py /tmp/ppf 24 ) > /root/repo/gpurun_out/prefill_${b}_a${a}.txt
done; done
