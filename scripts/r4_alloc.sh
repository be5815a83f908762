#!/bin/bash
# allocated vs reserved HBM of the config-B step: default streams, no side streams, expandable segments (the caching allocator keeps one block
# pool PER STREAM, and a block freed on one stream is not handed to another)
O=gpurun_out/alloc; mkdir -p $O
line() { tag=$1; shift; "$@" > $O/$tag.json 2> $O/$tag.err; python3 -c "
import json
d = json.loads([l for l in open('$O/$tag.json') if l.startswith('{')][-1]); c = d['config']
print('$tag', d['value'], d['ms_per_step'], 'allocated', c.get('peak_hbm_gb'), 'reserved', c.get('peak_hbm_reserved_gb'), flush=True)"; }
B="--steps 8 --warmup 3 --no-cpu-baseline --no-parity --roofline-steps 0"
line default python3 bench.py $B
line no_dw_stream env LECCR_DW_STREAM=0 python3 bench.py $B
line no_side_streams env LECCR_DW_STREAM=0 LECCR_TEXT_STREAM=0 LECCR_CAP_STREAM=0 python3 bench.py $B
line expandable env PYTORCH_HIP_ALLOC_CONF=expandable_segments:True python3 bench.py $B
line gc env PYTORCH_HIP_ALLOC_CONF=garbage_collection_threshold:0.6 python3 bench.py $B
