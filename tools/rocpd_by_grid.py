"""Per-(kernel, grid) durations from a rocprofv3 rocpd database: one line per distinct launch shape (= per layer) — which layer costs what.

    python tools/rocpd_by_grid.py gpurun_out/prof/bench_results.db [min_total_ms]
"""
import sqlite3
import sys


def main(path, min_ms=0.0):
    con = sqlite3.connect(path)
    rows = con.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, lds_size from kernels order by start").fetchall()
    agg = {}
    for name, s, e, gx, gy, gz, wx, vg, lds in rows:
        a = agg.setdefault((name, gx, gy, gz, wx), [0, 0, vg, lds])
        a[0] += 1
        a[1] += e - s
    tot = sum(a[1] for a in agg.values())
    print(f"# {path}: {len(rows)} dispatches, {tot / 1e6:.3f} ms of kernel time")
    print(f"{'calls':>6} {'total_ms':>9} {'avg_us':>9} {'%':>6} {'grid (threads)':>22} {'wg':>5} {'vgpr':>5} {'lds':>7}  name")
    for (name, gx, gy, gz, wx), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if a[1] / 1e6 < float(min_ms):
            continue
        print(f"{a[0]:6d} {a[1] / 1e6:9.3f} {a[1] / a[0] / 1e3:9.2f} {100 * a[1] / tot:6.2f} {f'{gx}x{gy}x{gz}':>22} {wx:5d} {a[2]:5d} {a[3]:7d}  {name[:120]}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
