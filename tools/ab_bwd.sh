export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | grep -v Warning | tail -6
