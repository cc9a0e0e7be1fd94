#!/bin/bash
for e in "LRG_FREE_RUN_WAVES=-1" "LRG_FREE_RUN_WAVES=1" "LRG_FREE_RUN_WAVES=1 LRG_ASYNC_RT_PARTS=1"; do
env $e timeout 400 python bench.py --gpus 1 --steps 8 --warmup 3 --cpu-seconds 0 --p0-rooms 0 --named-configs 0 --best-slots "" --steady-slots "" --one-room-ks "1,3" --fixed-rooms 0 --rooms 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=d['one_room_per_gpu']
print('$e', {k:{kk: round(vv['committed_steps_per_sec']) for kk,vv in o[k]['by_speculation_depth'].items()} for k in ('median_room','largest_room') if k in o})
"
done
