"""Summarise a rocprofv3 rocpd database (kernel trace): per kernel x grid, calls / median / min us.
    python tools/prof_db_summary.py <dir-or-db> [top N]"""
import collections, glob, os, sqlite3, sys
d = sys.argv[1]
f = d if d.endswith(".db") else sorted(glob.glob(d + "/**/*_results.db", recursive=True), key=os.path.getmtime)[-1]
cur = sqlite3.connect(f).cursor()
agg = collections.defaultdict(list)
for name, gx, wx, vg, lds, st, en in cur.execute(
        "select name, grid_x, workgroup_x, vgpr_count, lds_size, start, end from kernels"):
    short = name.split("(")[0].replace("void mi::", "")
    agg[(short, gx, wx, vg, lds)].append(en - st)
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
print(f"{'kernel':58s} {'grid':>8s} {'wg':>5s} {'vgpr':>5s} {'lds':>7s} {'calls':>6s} {'med_us':>8s} {'min_us':>8s} {'total_ms':>9s}")
for k, v in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    v = sorted(v)
    print(f"{k[0][:58]:58s} {k[1]:>8} {k[2]:>5} {k[3]:>5} {k[4]:>7} {len(v):6d} {v[len(v)//2]/1e3:8.2f} {v[0]/1e3:8.2f} {sum(v)/1e6:9.3f}")
