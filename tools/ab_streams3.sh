export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
B="--no-cpu-baseline --no-full-swap --no-pti --clip 0 --no-mask-sensitivity --soak-seconds 2 --no-in-run-ab"
for s in 2 3 4 2 3 4; do
  echo -n "streams=$s  "; timeout 300 python bench.py $B --streams $s 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',d['value'],'soak',d['soak_faces_per_s'],'one_stream',d['one_stream_faces_per_s'], 'W', d['soak_power_w'], 'MHz', d['soak_sclk_mhz'])"
done
