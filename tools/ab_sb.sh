export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_synthesis.py -m gpu -q --tb=short 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-swap 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['all_modconv3x3'])"
