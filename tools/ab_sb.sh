export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parser.py tests/test_gpu_encoder.py -m gpu -q --tb=short -x 2>&1 | tail -4
E4S_SWAP_TWO_STREAMS=0 timeout 300 python tools/time_swap.py 8 8 2>&1 | tail -6
timeout 300 python tools/time_swap.py 8 8 2>&1 | tail -5
