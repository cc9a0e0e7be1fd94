#!/bin/bash
# what faster pooled-product units could gain at many rooms in flight: --policy gt (the masks come from the ground truth, so builds with pieces compiled out stay comparable),
# units off (the library's choice above 176 slots) / sixteen units / sixteen units nobody waits for (-DLRG_EXP_NO_WAIT_POOLED=1: results wrong, dynamics unchanged)
mkdir -p gpurun_out
OUT=gpurun_out/r04_units_bound.txt
: > $OUT
run() {  # tag flags units slots
  LRG_HIPCC_FLAGS="$2" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
  for S in $4; do
  LRG_FREE_RUN_UNITS=$3 LRG_HIPCC_FLAGS="$2" timeout 600 python bench.py --gpus 1 --mode free --policy gt --rooms $S --steps 12 --warmup 4 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --fixed-rooms 0 > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - <<PY >> $OUT
import json
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print('%-28s slots %3d: %8.0f instance-steps/s  %.1f us/step/slot' % ('$1', $S, d['value'], d['us_per_instance_step_per_slot']))
PY
  done
}
SL="${SLOTS:-136 192 272}"
run "units off" "" -1 "$SL"
run "16 units" "" 16 "$SL"
run "16 units, nobody waits" "-DLRG_EXP_NO_WAIT_POOLED=1" 16 "$SL"
LRG_HIPCC_FLAGS="" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cat $OUT
