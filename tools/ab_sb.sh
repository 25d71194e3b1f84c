export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for v in 2 4 8; do echo "== E4S_SWAP_CHAINS=$v"; E4S_SWAP_CHAINS=$v timeout 300 python tools/time_swap.py 8 8 2>&1 | tail -5; done
