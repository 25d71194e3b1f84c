export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-swap > $R/gpurun_out/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_swap -o swap -- python3 $R/tools/time_swap.py 8 6 > $R/gpurun_out/prof_swap.log 2>&1
tail -7 $R/gpurun_out/prof_swap.log
