export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
python3 $R/tools/time_parts.py both 8 5; python3 $R/tools/time_parts.py both 16 5
for w in parse encode; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o $w -- python3 $R/tools/time_parts.py $w 8 6 > $R/gpurun_out/prof_$w.log 2>&1
python3 $R/tools/rocpd_by_grid.py $R/gpurun_out/prof_$w/${w}_results.db 0.02 > $R/gpurun_out/parts_${w}_kernel_stats.txt 2>&1 || true
done
ls $R/gpurun_out/prof_parse | head
