#!/bin/bash
# dev: SQ counter pass over tools/attn_prefill_time.py -> gpurun_out/pmc_attn_summary.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pa -o pa -- python3 /root/repo/tools/attn_prefill_time.py > /root/repo/gpurun_out/pmc_attn.log 2>&1
cd /root/repo
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pa/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = (row['Kernel_Name'][:48], row['Grid_Size'])
    agg[k][row['Counter_Name']] += float(row['Counter_Value'])
    if row['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[k] += 1
with open('gpurun_out/pmc_attn_summary.txt', 'w') as o:
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        if 'attn' not in k[0]: continue
        line = '%s grid=%s n=%d ' % (k[0], k[1], cnt[k]) + ' '.join('%s=%.4g' % (c, v / max(cnt[k], 1)) for c, v in sorted(d.items()))
        o.write(line + '\n'); print(line)
PY
