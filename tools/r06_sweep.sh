#!/bin/bash
# round 6: the steady leg of bench.py per environment in $1 (";"-separated); $2 = extra bench.py arguments (default: 68 rooms in flight)
IFS=';' read -ra ENVS <<< "${1:-X=1}"
for e in "${ENVS[@]}"; do
  env $e timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --one-room-ks "" --named-configs 0 --fixed-rooms 68 $2 2>gpurun_out/sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e: %.0f %s, %.1f us/step/slot, frac %.3f, fixed %s' % (d['value'], d['unit'], d['us_per_instance_step_per_slot'], d['roofline']['frac'], {k: d.get('fixed_work', {}).get(k) for k in ('rooms_per_sec', 'labels_crc32')}))
" || tail -5 gpurun_out/sweep.err
done
