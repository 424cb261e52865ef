#!/bin/bash
# dev: rocprofv3 kernel trace of tools/spec_profile.py <rows> <k>; summary -> gpurun_out/rows_summary.txt
set -e
ROWS=${1:-16}; K=${2:-4}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prows
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prows -o rows -- python3 /root/repo/tools/spec_profile.py $ROWS $K > /root/repo/gpurun_out/rows.log 2>&1
cd /root/repo
python3 tools/prof_summary.py /tmp/prows 24 > gpurun_out/rows_summary.txt
cat gpurun_out/rows_summary.txt
