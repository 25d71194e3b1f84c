export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pti -o pti -- python3 $R/tools/time_pti.py --steps 4 > $R/gpurun_out/prof_pti.log 2>&1
cd $R; python tools/rocpd_summary.py gpurun_out/prof_pti/pti_results.db 80 | cut -c1-230 > gpurun_out/pti_kernels.txt
rm -rf gpurun_out/prof_pti; tail -3 gpurun_out/prof_pti.log
