"""Dev tool: HBM traffic of the decode GEMVs from a rocprofv3 --pmc FETCH_SIZE pass of bench.py.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc -o bench -- \
        python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --ttft-prompts 1
    python tools/pmc_summary.py gpurun_out/pmc profiles/r01_gemv_traffic.json

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE is in KiB and on gfx950
reports exactly half of the bytes of a wide coalesced streaming read -> bytes = value * 1024 * 2.
"""
import collections, csv, glob, hashlib, json, os, sys

d, out = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE" or "gemv_" not in r["Kernel_Name"]:
        continue
    per[r["Kernel_Name"].split("(")[0].replace("void mi::", "")].append(float(r["Counter_Value"]))
n = sum(len(v) for v in per.values())
total = sum(sum(v) for v in per.values()) * 1024 * 2
doc = {
    "kernel": "mi::gemv_kernel / gemv_priv_kernel (all projections + lm_head of the decode steps in the pass)",
    "launches": n,
    "traffic_bytes_per_launch": round(total / n),
    "per_instantiation_median_bytes": {k: round(sorted(v)[len(v) // 2] * 2048) for k, v in per.items()},
    # the figure belongs to THESE kernels: bench.py flags it as stale when the source has changed since
    "kernel_source_sha1": hashlib.sha1(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                         "vllm-neuron_amd", "csrc", "linear_kernels.hip"), "rb").read()).hexdigest(),
    "counter": "FETCH_SIZE (KiB) x 1024 x 2: gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)",
    "command": "rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 4 --warmup 1 "
               "--no-cpu-baseline --ttft-prompts 1",
}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc, indent=1))
