export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for v in 0 8 16 24; do echo "== DBG=$v"; E4S_DBG=$v timeout 300 python tools/time_up.py 2>&1 | grep cin; done
