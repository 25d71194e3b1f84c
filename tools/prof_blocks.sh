export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
for m in half tophalf; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_blk_$m -o blk -- python3 $R/tools/time_blocks.py 8 $m > $R/gpurun_out/prof_blk_$m.log 2>&1
python3 $R/tools/rocpd_by_grid.py $R/gpurun_out/prof_blk_$m/blk_results.db 0.5 | grep -E "masked_up_block|4, 1, 1, 8, 5|uniform_blocks" | cut -c1-150
echo ---
done
