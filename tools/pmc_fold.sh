# PMC passes over the backward-part timing script (fold / dgrad kernels): counters + kernel trace only
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d $R/gpurun_out/pmc_fold_sq -o pmc -- python3 $R/tools/time_dgrad.py > $R/gpurun_out/pmc_fold_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fold_fetch -o pmc -- python3 $R/tools/time_dgrad.py > $R/gpurun_out/pmc_fold_fetch.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fold -o prof -- python3 $R/tools/time_dgrad.py > $R/gpurun_out/prof_fold.log 2>&1
cd $R
python tools/rocpd_pmc.py gpurun_out/pmc_fold_sq/pmc_results.db fold > gpurun_out/pmc_fold.txt
python tools/rocpd_pmc.py gpurun_out/pmc_fold_fetch/pmc_results.db fold >> gpurun_out/pmc_fold.txt
python tools/rocpd_by_grid.py gpurun_out/prof_fold/prof_results.db 2>/dev/null | grep -i "fold\|gemm_sb\|dgrad" | head -20 >> gpurun_out/pmc_fold.txt
rm -rf gpurun_out/pmc_fold_sq gpurun_out/pmc_fold_fetch gpurun_out/prof_fold
cat gpurun_out/pmc_fold.txt
