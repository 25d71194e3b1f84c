"""Per-kernel averages of the PMC counters in a rocprofv3 (rocpd sqlite) result.   python tools/rocpd_pmc.py file.db [name-filter]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [d[1] for d in cur.execute("pragma table_info('counters_collection')")]
rows = cur.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
agg = {}
for r in rows:
    name = r[ix["kernel_name"]] if "kernel_name" in ix else r[ix["name"]]
    if flt not in name: continue
    key = (name[:90], r[ix["counter_name"]])
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += r[ix["value"]]
kern = sorted({k[0] for k in agg})
for k in kern:
    print(k)
    for (kk, c), (n, v) in sorted(agg.items()):
        if kk == k: print(f"    {c:38s} calls={n:4d} avg={v/n:16.1f}")
