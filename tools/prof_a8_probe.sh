#!/bin/bash
# dev: kernel trace of tools/gemm_a8_probe.py for both tile kernels -> gpurun_out/a8_probe.txt
cd /tmp && export TMPDIR=/tmp
: > /root/repo/gpurun_out/a8_probe.txt
for m in 0 1; do
  rm -rf /tmp/pa8
  MI355X_A8_WIDE=$m timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pa8 -o a8 -- python3 /root/repo/tools/gemm_a8_probe.py "$@" > /root/repo/gpurun_out/a8_probe_$m.log 2>&1 || exit 1
  echo "== MI355X_A8_WIDE=$m" >> /root/repo/gpurun_out/a8_probe.txt
  python3 /root/repo/tools/prof_summary.py /tmp/pa8 40 | grep "gemm_a8\|kernel " >> /root/repo/gpurun_out/a8_probe.txt
done
cat /root/repo/gpurun_out/a8_probe.txt
