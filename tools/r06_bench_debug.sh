#!/bin/bash
# tools/r03_bench_debug.sh with the number of rooms in flight as $2: the stage-by-stage breakdown (-DLRG_ASYNC_DEBUG=1) of the steady leg, per environment in $1 (";"-separated)
mkdir -p gpurun_out
ROOMS=${2:-400}
export LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="$LRG_HIPCC_FLAGS -DLRG_ASYNC_DEBUG=1"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
IFS=';' read -ra ENVS <<< "${1:-X=1}"
for e in "${ENVS[@]}"; do
  env $e timeout 600 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-5} --rooms $ROOMS --cpu-seconds 0 --p0-rooms 0 --named-configs 0 --best-slots "" --steady-slots "" --fixed-rooms 0 --one-room-ks "" 2> gpurun_out/bench_dbg.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e rooms $ROOMS: %.0f %s, %.1f us/step/slot' % (d['value'], d['unit'], d['us_per_instance_step_per_slot']))
" | tee -a gpurun_out/r06_bench_debug_$ROOMS.log
  grep '^{' gpurun_out/bench_dbg.err | tee -a gpurun_out/r06_bench_debug_$ROOMS.log
done
