export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_synthesis.py -m gpu -q -x --tb=short 2>&1 | grep -v Warning | tail -2
for e in 0 1; do echo "exp $e"; E4S_UF_EXP=$e E4S_HIP_LIB=$R/e4s2024_amd/lib/libe4s_hip_prof.so timeout 600 python tools/phase_prof.py 2>&1 | grep -A9 "fused up 64" | grep "kernel\|K loop\|epilogue compute"; done
timeout 600 python bench.py --no-cpu-baseline --no-pti --no-full-swap 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['roofline']['all_modconv3x3']['by_kernel_ms_per_step']; print(d['value'], d['ms_per_step'], 'upfused', k['modconv_up_fused_sb'])"
