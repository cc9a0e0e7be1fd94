#!/bin/bash
# the steady leg of bench.py in a -DLRG_ASYNC_DEBUG=1 build: the stage-by-stage breakdown of the timed launches (stderr of bench.py), per environment in $1 (";"-separated)
mkdir -p gpurun_out
export LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="$LRG_HIPCC_FLAGS -DLRG_ASYNC_DEBUG=1"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
IFS=';' read -ra ENVS <<< "${1:-X=1}"
for e in "${ENVS[@]}"; do
  env $e timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --named-configs 0 --best-slots "" --steady-slots "" --fixed-rooms 68 2> gpurun_out/bench_dbg.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e: %.0f %s, %.1f us/step/slot' % (d['value'], d['unit'], d['us_per_instance_step_per_slot']))
" | tee -a gpurun_out/r03_bench_debug.log
  grep '^{' gpurun_out/bench_dbg.err | tee -a gpurun_out/r03_bench_debug.log
done
