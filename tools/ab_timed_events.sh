# what do the dominant kernel's events inside the timed region, and a cold start of the timed region, cost the headline?  (round 5)
R=$GRAFT_REPO_ROOT; cd $R
F="--steps 20 --warmup 5 --no-cpu-baseline --no-in-run-ab --no-full-swap --no-pti --clip 0 --no-mask-sensitivity"
for rep in 1 2 3; do
  for cfg in "--timed-events dominant" "--timed-events none" "--timed-events dominant --settle-seconds 2" "--timed-events none --settle-seconds 2"; do
    echo "== $cfg"
    timeout 300 python bench.py $F $cfg 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'soak', d['soak']['faces_per_s'], 'one_stream', d['one_stream']['faces_per_s'])"
  done
done
