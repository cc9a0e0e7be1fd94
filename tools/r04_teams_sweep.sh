#!/bin/bash
# tile teams per worker workgroup at many rooms in flight (free-running steady leg of bench.py)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
OUT=gpurun_out/r04_teams_sweep.txt
: > $OUT
for S in ${SLOTS:-192 136 272 68}; do
for T in ${TEAMS:-2 3 4}; do
  LRG_FREE_RUN_TEAMS=$T timeout 600 python bench.py --gpus 1 --mode free --rooms $S --steps 12 --warmup 4 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --fixed-rooms 0 > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - <<PY >> $OUT
import json
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print('slots %3d teams %d: %8.0f instance-steps/s  %.1f us/step/slot  frac %.3f' % ($S, $T, d['value'], d['us_per_instance_step_per_slot'], d['roofline']['frac']))
PY
done; done
cat $OUT
