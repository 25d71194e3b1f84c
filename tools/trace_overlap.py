"""How much do the kernels of the two alternating streams of the headline bench really overlap?  Reads a rocprofv3 kernel trace (rocpd sqlite) and prints, for the
steady-state part, (i) total kernel time, (ii) wall time covered by at least one kernel, (iii) time covered by two or more, and per kernel name the share of its
own time during which another kernel was running."""
import sqlite3, sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, lds_size, stream_id from kernels order by start").fetchall() if True else []
if not rows:
    sys.exit("no kernels")
# the window: [t0, t0 + win) where t0 = the start of the `skip`-th launch of the dominant masked kernel (argv[3], default 40: past the warm-up)
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 30e6
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dom = [r for r in rows if "region_modconv_mx_kernel" in r[0]]
t0 = dom[min(skip, len(dom) - 1)][1] if dom else rows[0][1]
rows = [r for r in rows if t0 <= r[1] < t0 + win]
ev = []
for i, (n, s, e, lds, st) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set(); last = ev[0][0]; cov1 = cov2 = 0
shared = defaultdict(float); own = defaultdict(float)
for t, d, i in ev:
    dt = t - last
    if active:
        cov1 += dt
        if len(active) > 1:
            cov2 += dt
        for j in active:
            own[rows[j][0]] += dt
            if len(active) > 1:
                shared[rows[j][0]] += dt
    last = t
    if d > 0: active.add(i)
    else: active.discard(i)
tot = sum(e - s for _, s, e, _, _ in rows)
span = rows[-1][2] - rows[0][1]
print(f"window {span/1e6:.2f} ms: sum of kernel time {tot/1e6:.2f} ms, covered by >= 1 kernel {cov1/1e6:.2f} ms, by >= 2 kernels {cov2/1e6:.2f} ms, idle {(span-cov1)/1e6:.2f} ms; streams {sorted(set(r[4] for r in rows))}")
for n, o in sorted(own.items(), key=lambda kv: -kv[1])[:22]:
    print(f"  {o/1e6:8.3f} ms own, {100*shared[n]/o:5.1f} % of it beside another kernel   {n[:110]}")
