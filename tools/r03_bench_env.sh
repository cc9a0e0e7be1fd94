#!/bin/bash
# the driver's bench line under a list of environments ("VAR=val VAR2=val;VAR=val2;..."), short: no cpu leg, no sweep
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
IFS=';' read -ra ENVS <<< "$1"
for e in "${ENVS[@]}"; do
  env $e timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" 2> gpurun_out/bench_env.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$e: %.0f %s, %.0f rooms/s, %.1f us/step/slot, roofline %.3f, given_up %s' % (d['value'], d['unit'], d['rooms_per_sec'], d['us_per_instance_step_per_slot'], r['frac'], d['fixed_work'].get('given_up')))
" | tee -a gpurun_out/r03_bench_env.log
done
