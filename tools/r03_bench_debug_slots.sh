#!/bin/bash
# stage breakdown of the free-running launches (debug build) at a given number of rooms in flight: $1 = slots, $2 = env (optional)
mkdir -p gpurun_out
export LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="$LRG_HIPCC_FLAGS -DLRG_ASYNC_DEBUG=1"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
env ${2:-X=1} timeout 600 python bench.py --gpus 1 --steps 10 --warmup 4 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --fixed-rooms 0 --rooms $1 --mode free 2> gpurun_out/bench_dbg.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 slots ${2}: %.0f %s, %.1f us/step/slot' % (d['value'], d['unit'], d['us_per_instance_step_per_slot']))
" | tee -a gpurun_out/r03_bench_debug_slots.log
grep '^{' gpurun_out/bench_dbg.err | tee -a gpurun_out/r03_bench_debug_slots.log
