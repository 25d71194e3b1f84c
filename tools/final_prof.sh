# Round evidence: GPU test suite, default bench line, kernel-trace stats of the bench / full swap / PTI, three PMC passes of the bench.
# Everything lands as small text / JSON under gpurun_out/ (copy what is to be judged into profiles/); the result databases are deleted on the box.
# usage: bash tools/final_prof.sh <tag>      (e.g. r06_final)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=${1:-r06_final}; mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${T}_gpu_tests.txt; cat gpurun_out/${T}_gpu_tests.txt
cp gpurun_out/parity.json gpurun_out/${T}_parity.json 2>/dev/null
timeout 900 python bench.py 2> gpurun_out/${T}_bench.err | grep '^{' > gpurun_out/${T}_bench.json; cut -c1-300 gpurun_out/${T}_bench.json
cp gpurun_out/bench_detail.json gpurun_out/${T}_bench_detail.json 2>/dev/null      # (the compact line is what the driver sees; the detail object carries by_layer, in_run_ab ...)
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-in-run-ab --no-full-swap --no-pti --clip 0 --soak-seconds 0 > $R/gpurun_out/prof_bench.log 2>&1
grep '^{' $R/gpurun_out/prof_bench.log > $R/gpurun_out/${T}_bench_under_rocprof.json
# the same on one stream: the kernel durations the bench's roofline object is computed from (under the default two-stream overlap a kernel shares the chip)
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench1 -o bench -- python3 $R/bench.py --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-in-run-ab --no-full-swap --no-pti --clip 0 --no-mask-sensitivity --soak-seconds 0 > $R/gpurun_out/prof_bench1.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_swap -o swap -- python3 $R/tools/time_swap.py 8 6 > $R/gpurun_out/prof_swap.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_bench/bench_results.db | cut -c1-260 > gpurun_out/${T}_bench_kernel_stats.txt
python tools/rocpd_by_grid.py gpurun_out/prof_bench/bench_results.db 0.02 | cut -c1-260 > gpurun_out/${T}_bench_by_layer.txt
python tools/rocpd_summary.py gpurun_out/prof_bench1/bench_results.db | cut -c1-260 > gpurun_out/${T}_bench_one_stream_kernel_stats.txt
python tools/rocpd_by_grid.py gpurun_out/prof_bench1/bench_results.db 0.02 | cut -c1-260 > gpurun_out/${T}_bench_one_stream_by_layer.txt
rm -rf gpurun_out/prof_bench1
python tools/rocpd_summary.py gpurun_out/prof_swap/swap_results.db | cut -c1-260 > gpurun_out/${T}_swap_kernel_stats.txt
grep -E "ms / batch|ms per face" gpurun_out/prof_swap.log > gpurun_out/${T}_swap_timing.txt
rm -rf gpurun_out/prof_bench gpurun_out/prof_swap
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pti -o pti -- python3 $R/tools/time_pti.py --steps 4 > $R/gpurun_out/prof_pti.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_pti/pti_results.db 60 | cut -c1-260 > gpurun_out/${T}_pti_kernel_stats.txt   # steady state: the last 60 ms = graph replays
grep "PTI step" gpurun_out/prof_pti.log > gpurun_out/${T}_pti_timing.txt
rm -rf gpurun_out/prof_pti
bash tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" pmc_sq
cd $R; python tools/rocpd_pmc.py gpurun_out/pmc_sq/pmc_results.db region_modconv > gpurun_out/${T}_pmc_sq.txt; python tools/rocpd_pmc.py gpurun_out/pmc_sq/pmc_results.db region_upconv >> gpurun_out/${T}_pmc_sq.txt; python tools/rocpd_pmc.py gpurun_out/pmc_sq/pmc_results.db up_hc >> gpurun_out/${T}_pmc_sq.txt; python tools/rocpd_pmc.py gpurun_out/pmc_sq/pmc_results.db masked_up_block >> gpurun_out/${T}_pmc_sq.txt
python tools/rocpd_pmc.py gpurun_out/pmc_sq/pmc_results.db chain_conv >> gpurun_out/${T}_pmc_sq.txt; rm -rf gpurun_out/pmc_sq
bash tools/pmc_pass.sh "FETCH_SIZE" pmc_fetch
cd $R; python tools/rocpd_pmc.py gpurun_out/pmc_fetch/pmc_results.db > gpurun_out/${T}_pmc_fetch.txt; rm -rf gpurun_out/pmc_fetch
bash tools/pmc_pass.sh "WRITE_SIZE" pmc_write
cd $R; python tools/rocpd_pmc.py gpurun_out/pmc_write/pmc_results.db > gpurun_out/${T}_pmc_write.txt; rm -rf gpurun_out/pmc_write
head -14 gpurun_out/${T}_bench_kernel_stats.txt; head -30 gpurun_out/${T}_pmc_sq.txt
