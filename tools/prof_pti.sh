export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pti -o pti -- python3 $R/tools/time_pti.py --steps 4 > $R/gpurun_out/prof_pti.log 2>&1
tail -3 $R/gpurun_out/prof_pti.log
python3 $R/tools/rocpd_summary.py $R/gpurun_out/prof_pti/pti_results.db 45 | cut -c1-200 > $R/gpurun_out/pti_kernel_stats.txt
head -45 $R/gpurun_out/pti_kernel_stats.txt
