#!/bin/bash
# the steady leg of bench.py in a -DLRG_ASYNC_DEBUG=1 build: stage-by-stage breakdown of the timed launches, per argument set in $1 (";"-separated bench.py arguments)
mkdir -p gpurun_out
export LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="$LRG_HIPCC_FLAGS -DLRG_ASYNC_DEBUG=1"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
OUT=gpurun_out/${2:-r05_bench_debug}.log
: > $OUT
IFS=';' read -ra SETS <<< "${1:---rooms 68}"
for a in "${SETS[@]}"; do
  timeout 600 python bench.py --gpus 1 --steps 12 --warmup 4 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --fixed-rooms 0 --one-room-ks= $a 2> gpurun_out/bench_dbg.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$a: %.0f %s, %.1f us/step/slot, roofline %.3f' % (d['value'], d['unit'], d['us_per_instance_step_per_slot'], d['roofline']['frac']))
" | tee -a $OUT
  grep '^{' gpurun_out/bench_dbg.err | tee -a $OUT
done
