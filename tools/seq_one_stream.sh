# launch sequence of one steady-state one-stream step (names, grids, durations, gaps) -> gpurun_out/${1}_step_sequence.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
T=${1:-r06}
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_seq -o bench -- python3 $R/bench.py --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-in-run-ab --no-full-swap --no-pti --clip 0 --no-mask-sensitivity --soak-seconds 0 > $R/gpurun_out/prof_seq.log 2>&1
cd $R
python tools/trace_sequence.py gpurun_out/prof_seq/bench_results.db 8 | cut -c1-180 > gpurun_out/${T}_step_sequence.txt
rm -rf gpurun_out/prof_seq
cat gpurun_out/${T}_step_sequence.txt
