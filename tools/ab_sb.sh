export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q --tb=short -x 2>&1 | tail -3
for v in 0 1; do echo "== E4S_CONV_WIDE=$v"; E4S_SWAP_TWO_STREAMS=0 E4S_CONV_WIDE=$v timeout 300 python tools/time_swap.py 8 8 2>&1 | grep "encode_x2\|total"; E4S_CONV_WIDE=$v timeout 300 python tools/time_swap.py 8 8 2>&1 | grep "total"; done
