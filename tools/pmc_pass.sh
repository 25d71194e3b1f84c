# one PMC pass over the bench (counters + kernel trace only; no other trace domains)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace --pmc $1 -d $R/gpurun_out/$2 -o pmc -- python3 $R/bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-in-run-ab --no-full-swap --no-pti --clip 0 --no-mask-sensitivity --soak-seconds 0 > $R/gpurun_out/$2.log 2>&1
tail -2 $R/gpurun_out/$2.log | cut -c1-300
