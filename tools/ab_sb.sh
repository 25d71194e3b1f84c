export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_parser.py -m gpu -q --tb=short -x 2>&1 | tail -4
for v in 1 0; do echo "== E4S_CONV_BIGTILE=$v"; E4S_CONV_BIGTILE=$v timeout 300 python tools/time_conv.py 8 2>&1 | tail -13; E4S_CONV_BIGTILE=$v timeout 300 python tools/time_swap.py 8 6 2>&1 | grep "encode_x2\|parse_x2\|total"; done
