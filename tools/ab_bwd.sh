export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_backward.py -m gpu -q -x --tb=line 2>&1 | grep -v Warning | grep "Error\|assert\|passed\|failed" | cut -c1-300 | head
timeout 600 python tools/time_bwd_parts.py 2>&1 | tail -7 | cut -c1-200
timeout 600 python tools/time_pti.py --steps 6 2>&1 | tail -2
