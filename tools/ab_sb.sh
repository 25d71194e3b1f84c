export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python tools/time_pti.py --steps 3 2>&1 | tail -3
