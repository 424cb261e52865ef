"""Summarise a rocprofv3 kernel trace CSV: per kernel name x grid, count / median / min us."""
import collections, csv, glob, sys
d = sys.argv[1]
import os
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("void mi::", "")
    agg[(short, r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["LDS_Block_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
print(f"{'kernel':46s} {'grid':>8s} {'wg':>5s} {'vgpr':>5s} {'lds':>7s} {'calls':>6s} {'med_us':>8s} {'min_us':>8s} {'total_ms':>9s}")
for k, v in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    v = sorted(v)
    print(f"{k[0][:46]:46s} {k[1]:>8s} {k[2]:>5s} {k[3]:>5s} {k[4]:>7s} {len(v):6d} {v[len(v)//2]/1e3:8.2f} {v[0]/1e3:8.2f} {sum(v)/1e6:9.3f}")
