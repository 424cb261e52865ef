#!/bin/bash
# dev: rocprofv3 kernel trace of tools/decode_ab.py <args>; summary -> gpurun_out/decode_summary.txt
set -e
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pdec
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pdec -o dec -- python3 /root/repo/tools/decode_ab.py "$@" > /root/repo/gpurun_out/decode.log 2>&1
cd /root/repo
python3 tools/prof_summary.py /tmp/pdec 12 > gpurun_out/decode_summary.txt
tail -2 gpurun_out/decode.log
cat gpurun_out/decode_summary.txt
