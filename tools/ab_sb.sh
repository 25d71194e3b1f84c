export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_synthesis.py -m gpu -q --tb=short -k ragged 2>&1 | tail -25
