# launch sequence of ONE steady-state full-swap batch (names, grids, durations, gaps, stream) -> gpurun_out/${1}_swap_sequence.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
T=${1:-r06}
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_sseq -o swap -- python3 $R/tools/time_swap.py 8 4 > $R/gpurun_out/prof_sseq.log 2>&1
cd $R
python - <<'PY' > gpurun_out/${T}_swap_sequence.txt
import sqlite3
con = sqlite3.connect("gpurun_out/prof_sseq/swap_results.db")
rows = con.execute("select name, start, end, grid_x, stream_id from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if r[0].startswith("tensor2im")]
a, b = marks[-2] + 1, marks[-1] + 1                      # the last batch: behind the previous batch's tensor2im up to its own
t0 = rows[a][1]; prev = {}
print(f"# one full-swap batch of 8: {b - a} launches, {(rows[b - 1][2] - t0) / 1e6:.3f} ms first start .. last end")
for n, s, e, g, st in rows[a:b]:
    gap = (s - prev[st]) / 1e3 if st in prev else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  grid {g:8d}  st {st}  {n[:110]}")
    prev[st] = e
PY
rm -rf gpurun_out/prof_sseq
head -200 gpurun_out/${T}_swap_sequence.txt | cut -c1-170
