# Board power / shader clock while the headline step runs (a 12 s soak of bench.py's step on two streams), sampled twice a second with rocm-smi.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
B="--no-cpu-baseline --no-full-swap --no-pti --clip 0 --no-mask-sensitivity --no-in-run-ab --soak-seconds ${1:-12} ${2:-}"
echo "idle:"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | head -4
python bench.py $B > gpurun_out/power_probe_bench.json 2>/dev/null &
BP=$!
sleep 9
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed -e 's/.*: //' | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
python -c "
import json;d=json.loads(open('gpurun_out/power_probe_bench.json').read().strip().splitlines()[-1]);print('value',d['value'],'soak',d['soak_faces_per_s'])"
