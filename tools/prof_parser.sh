# kernel table of the face parser alone (tools/time_parts.py parse 16) -> gpurun_out/${1}_parser_kernel_stats.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_par -o par -- python3 $R/tools/time_parts.py parse 16 10 > $R/gpurun_out/prof_par.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_par/par_results.db | cut -c1-200 > gpurun_out/${1:-r04}_parser_kernel_stats.txt
rm -rf gpurun_out/prof_par
tail -1 gpurun_out/prof_par.log
