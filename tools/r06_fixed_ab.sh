#!/bin/bash
# round 6: the fixed-work leg (2 176 room jobs) at the slot counts in $2 (near-equal counts = repetitions within one process), per environment in $1 (";"-separated)
IFS=';' read -ra ENVS <<< "${1:-X=1}"
for e in "${ENVS[@]}"; do
  env $e timeout 600 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --named-configs 0 --one-room-ks "" --steady-slots "" --best-slots "${2:-396,400,404}" 2>gpurun_out/fixed_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=d['fixed_work_best']
print('$e:', {k:(round(v['rooms_per_sec'],1),round(v['instance_steps_per_sec']/1e3),v['formulation'][:4]) for k,v in b['sweep'].items()}, b['labels_crc32'])
" || tail -5 gpurun_out/fixed_ab.err
done
