# one-stream kernel table of the headline step (what each launch costs with the chip to itself) -> gpurun_out/${1}_one_stream_kernel_stats.txt, then the default bench line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
T=${1:-r04_mid}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench1 -o bench -- python3 $R/bench.py --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-in-run-ab --no-full-swap --no-pti --clip 0 --no-mask-sensitivity --soak-seconds 0 > $R/gpurun_out/prof_bench1.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_bench1/bench_results.db | cut -c1-200 > gpurun_out/${T}_one_stream_kernel_stats.txt
rm -rf gpurun_out/prof_bench1
python bench.py --no-cpu-baseline --no-full-swap --no-pti --clip 0 --no-mask-sensitivity > gpurun_out/${T}_bench_quick.json 2> gpurun_out/${T}_bench_quick.err
head -c 400 gpurun_out/${T}_bench_quick.json
