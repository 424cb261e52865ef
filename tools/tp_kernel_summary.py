"""Dev tool: per-rank kernel time of a token-generation step from a rocprofv3 kernel trace of
`bench.py --gpus T --tp-loopback` (every rank shard on one GPU, one stream): the trace holds the
kernels of all T ranks back to back, so (sum of durations of the step's kernels) / T is the compute
one GPU would spend per step with a free exchange, and the ar_* rows are the exchange kernels'
single-GPU cost.    python tools/tp_kernel_summary.py <trace dir> T"""
import collections, csv, glob, json, os, sys
d, T = sys.argv[1], int(sys.argv[2])
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = [(r["Kernel_Name"].split("(")[0].replace("void mi::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
        for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
# the timed replay region = the last K steps: find decode steps as runs delimited by embed_kernel launches of 4 rows...
agg = collections.defaultdict(list)
for n, s, e in rows:
    agg[n].append(e - s)
out = {}
for k, v in agg.items():
    if any(x in k for x in ("ar_", "gemv", "attn_decode", "attn_combine")):
        v = sorted(v)
        out[k] = {"calls": len(v), "median_us": round(v[len(v) // 2] / 1e3, 2)}
print(json.dumps({"tp": T, "kernels": out}, indent=1))
