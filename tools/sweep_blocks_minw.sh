for lab in portrait coarse blocky; do for cfg in "0 128" "1 32" "1 64" "1 128"; do set -- $cfg; for rep in 1 2; do
E4S_UP_BLOCKS=$1 E4S_UP_BLOCKS_MINW=$2 python bench.py --labels $lab --no-cpu-baseline --no-pti --clip 0 --no-full-swap --no-mask-sensitivity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lab', '$1', '$2', d['value'], d['ms_per_step'])"; done; done; done
