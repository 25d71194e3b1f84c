# the GPU suite's core files under each documented route switch: the non-default routes must stay green
R=$GRAFT_REPO_ROOT; cd $R
SW=${1:-E4S_ENC_PREP=0 E4S_ENC_C4=0,E4S_ENC_PREP=0 E4S_MX=0 E4S_MX=1 E4S_UP_HC=0 E4S_SP_CHAIN=0 E4S_UP_BLOCKS=0 E4S_MX3=0 E4S_WINOGRAD=0 E4S_ENC_ROUTE_BY_IMAGE=1 E4S_MODCONV=f32}
for sw in $SW; do
  echo "== $sw"; env ${sw//,/ } python -m pytest tests/test_gpu_synthesis.py tests/test_gpu_encoder.py tests/test_gpu_pipeline.py tests/test_gpu_chain.py tests/test_gpu_upblock_mx.py tests/test_gpu_mx4.py tests/test_gpu_f16_guard.py -q -m gpu 2>&1 | grep -E "^FAILED|^E  |passed|failed" | cut -c1-250 | head -12
done
