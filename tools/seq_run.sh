export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_seq -o seq -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-in-run-ab --no-full-swap --no-pti --clip 0 --streams 1 --no-mask-sensitivity --soak-seconds 0 > $R/gpurun_out/prof_seq.log 2>&1
cd $R
python tools/trace_sequence.py gpurun_out/prof_seq/seq_results.db 8 > gpurun_out/seq.txt 2>&1
rm -rf gpurun_out/prof_seq
tail -3 gpurun_out/prof_seq.log | cut -c1-300
