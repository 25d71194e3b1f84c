export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_backward.py -m gpu -q -x --tb=short -k "version_counter" 2>&1 | grep -v Warning | tail -15
