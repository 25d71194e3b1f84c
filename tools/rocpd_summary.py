"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the text table committed under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof1/bench_results.db > profiles/r01_xxx_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, last_ms=None):
    """``last_ms``: only the dispatches that started in the last ``last_ms`` milliseconds of the trace (e.g. the steady-state replays of a
    script whose warm-up ran MIOpen's solver search)."""
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, accum_vgpr_count, lds_size "
                       "from kernels order by start").fetchall()
    if last_ms is not None and rows:
        t0 = rows[-1][2] - float(last_ms) * 1e6
        rows = [r for r in rows if r[1] >= t0]
    agg = {}
    for name, s, e, gx, gy, gz, wx, vg, ag, lds in rows:
        a = agg.setdefault(name, [0, 0, 10**18, 0, vg, ag, lds])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    span = rows[-1][2] - rows[0][1] if rows else 0
    print(f"# {path}: {len(rows)} dispatches, {len(agg)} kernels, sum of kernel time {tot/1e6:.3f} ms, first-start..last-end {span/1e6:.3f} ms")
    print(f"{'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'%':>6} {'vgpr':>5} {'agpr':>5} {'lds':>6}  name")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{a[0]:6d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:10.2f} {a[2]/1e3:9.2f} {a[3]/1e3:9.2f} {100*a[1]/tot:6.2f} {a[4]:5d} {a[5]:5d} {a[6]:6d}  {name[:150]}")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))
