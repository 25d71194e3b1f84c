# per-layer one-stream times of the masked layers under each kind of region map (bench.py --labels ...): which layers pay for the map
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
B="--no-cpu-baseline --no-full-swap --no-pti --clip 0 --no-mask-sensitivity --soak-seconds 0 --no-in-run-ab"
for lab in blocky portrait coarse iid; do
  echo "== $lab"; timeout 300 python bench.py $B --labels $lab 2>&1 >/dev/null | grep "^bench detail: " | cut -c15- | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value',d['value'],'one_stream',d['one_stream']['faces_per_s'])
for row in r['by_layer']: print('   %-20s %.4f ms  exec/alg %s  uniform %s' % (row['layer'], row['ms_per_step'], row['executed_over_algorithmic'], row.get('uniform_block_share')))
print('  ', r['all_modconv3x3']['by_kernel_ms_per_step'])"
done
