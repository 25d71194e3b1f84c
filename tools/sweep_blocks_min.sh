# the share of qualifying tiles from which a layer uses the block path, on portrait-shaped maps (one stream, so that kernel times add up)
for m in 100 40 20 5; do E4S_UP_BLOCKS_MIN=$m python bench.py --streams 1 --labels portrait --no-cpu-baseline --no-pti --clip 0 --no-full-swap --no-mask-sensitivity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('min $m', d['value'], d['ms_per_step'])"; done
python tools/time_blocks.py 8 2>&1 | tail -4 | cut -c1-200
E4S_UP_BLOCKS_MIN=5 python tools/time_blocks.py 8 2>&1 | tail -2 | cut -c1-200
