export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_maskops.py -m gpu -q -x --tb=short 2>&1 | grep -v Warning | tail -12
