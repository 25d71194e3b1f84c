# A/B of the persistent chain kernels' tile walk (E4S_WALK: XCD-aware offsets + band-blocked enumeration; E4S_HC_REV: up layers last-to-first) on one box:
# the compact bench line's value / one-stream / stage_ms per setting.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
B="--no-cpu-baseline --no-full-swap --no-pti --clip 0 --no-mask-sensitivity --soak-seconds 0 --no-in-run-ab"
for cfg in "0 0" "0 1" "1 0" "2049 0" "2049 1" "1025 1" "4097 1" "0 0" "2049 1"; do
  set -- $cfg
  echo "== E4S_WALK=$1 E4S_HC_REV=$2"
  E4S_WALK=$1 E4S_HC_REV=$2 timeout 300 python bench.py $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],'one_stream',d['one_stream_faces_per_s'],'stage',d['stage_ms'])"
done
