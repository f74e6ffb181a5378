"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max (us).

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/xxx_kernel_stats.md
"""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                 "max(grid_size), max(workgroup_size), max(lds_size), max(scratch_size) from "
                 "(select k.name as name, k.start as start, k.end as end, k.grid_size as grid_size, k.workgroup_size as workgroup_size,"
                 " k.lds_size as lds_size, k.scratch_size as scratch_size from kernels k) group by name order by 3 desc").fetchall() \
    if False else None
cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
sel = "name, start, end"
extra = [x for x in ("grid_x", "grid_size", "workgroup_x", "workgroup_size", "lds_size", "scratch_size", "vgpr_count", "sgpr_count") if x in cols]
q = f"select {sel}{''.join(', ' + e for e in extra)} from kernels"
data = c.execute(q).fetchall()
agg = {}
for r in data:
    name, s, e = r[0], r[1], r[2]
    a = agg.setdefault(name, {"n": 0, "tot": 0, "min": 1 << 62, "max": 0, "extra": r[3:]})
    d = e - s
    a["n"] += 1
    a["tot"] += d
    a["min"] = min(a["min"], d)
    a["max"] = max(a["max"], d)
total = sum(a["tot"] for a in agg.values()) or 1
print(f"# rocprofv3 --kernel-trace summary of `{db}`\n")
print("| kernel | calls | total ms | avg us | min us | max us | % | " + " | ".join(extra) + " |")
print("|---|---|---|---|---|---|---|" + "---|" * len(extra))
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
    print(f"| `{name[:110]}` | {a['n']} | {a['tot']/1e6:.3f} | {a['tot']/a['n']/1e3:.2f} | {a['min']/1e3:.2f} | {a['max']/1e3:.2f} | "
          f"{100*a['tot']/total:.1f} | " + " | ".join(str(x) for x in a["extra"]) + " |")
