export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_synthesis.py -m gpu -q -x --tb=short 2>&1 | grep -v Warning | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-pti --no-full-swap 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['roofline']['all_modconv3x3']['by_kernel_ms_per_step']; print(d['value'], d['ms_per_step'], 'upfused', k['modconv_up_fused_sb'])"
