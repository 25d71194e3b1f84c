R=$GRAFT_REPO_ROOT; cd $R
F="--steps 20 --warmup 5 --no-cpu-baseline --no-in-run-ab --no-full-swap --no-pti --clip 0 --no-mask-sensitivity"
for rep in 1 2 3; do
  for v in 0 1; do
    for lab in blocky portrait; do
    echo "== E4S_MXE=$v labels=$lab"
    E4S_MXE=$v timeout 300 python bench.py $F --labels $lab 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'soak', d['soak']['faces_per_s'], 'one_stream', d['one_stream']['faces_per_s'])"
    done
  done
done
