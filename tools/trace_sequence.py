"""Print the launch sequence (name, grid, duration) of one steady-state step from a rocprofv3 kernel trace (rocpd sqlite): the launches between the
`skip`-th and `skip + 1`-th occurrence of the step's first kernel (onehot_to_labels)."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, grid_x, stream_id from kernels order by start").fetchall()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
marks = [i for i, r in enumerate(rows) if "onehot_to_labels" in r[0]]
a, b = marks[skip], marks[skip + 1]
prev = rows[a][1]
for n, s, e, g, st in rows[a:b]:
    print(f"{(e - s) / 1e3:8.1f} us  gap_before {(s - prev) / 1e3:6.1f}  grid {g:8d}  stream {st}  {n[:100]}")
    prev = e
