# overlapped (2 / 3 / 4 streams) against one stream, in one session on one box
for s in 2 3 4 2 3 4; do python bench.py --streams $s --no-cpu-baseline --no-pti --clip 0 --no-full-swap --no-mask-sensitivity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $s steps', d['steps'], 'overlapped', d['value'], d['ms_per_step'], 'one', d['one_stream']['faces_per_s'], 'equal', d['one_stream']['images_equal_overlapped'])"; done
