export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_synthesis.py -m gpu -q --tb=short -x 2>&1 | tail -3
timeout 300 python tools/time_up.py 2>&1 | grep cin
for rep in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-swap 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['all_modconv3x3']['by_kernel_ms_per_step'])"; done
