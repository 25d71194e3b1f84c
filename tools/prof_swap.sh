# kernel table of the full swap (tools/time_swap.py: batch 8, 6 timed batches) -> gpurun_out/${1}_swap_kernel_stats.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_swap -o swap -- python3 $R/tools/time_swap.py 8 6 > $R/gpurun_out/prof_swap.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_swap/swap_results.db | cut -c1-220 > gpurun_out/${1:-r04_mid}_swap_kernel_stats.txt
rm -rf gpurun_out/prof_swap
tail -6 gpurun_out/prof_swap.log
