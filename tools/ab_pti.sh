export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python tools/time_pti.py --steps 4 2>&1 | tail -34
