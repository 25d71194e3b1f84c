export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/bench_vmcnt.json
python -c "
import json
d=json.load(open('gpurun_out/bench_vmcnt.json')); print(d['value'], d['ms_per_step'], d['roofline']['all_modconv3x3']['by_kernel_ms_per_step']); print(d.get('full_swap'))"
