# kernel-trace profile of the headline bench (short), per-kernel and per-(kernel, grid) summaries under gpurun_out/
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-in-run-ab --no-full-swap --no-pti --clip 0 > $R/gpurun_out/prof_bench.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_bench/bench_results.db | cut -c1-230 > gpurun_out/${1:-r02}_bench_kernel_stats.txt
python tools/rocpd_by_grid.py gpurun_out/prof_bench/bench_results.db 0.05 | cut -c1-230 > gpurun_out/${1:-r02}_bench_by_layer.txt
grep '^{' gpurun_out/prof_bench.log | cut -c1-300
rm -rf gpurun_out/prof_bench
head -30 gpurun_out/${1:-r02}_bench_by_layer.txt
