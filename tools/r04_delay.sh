#!/bin/bash
# Stage weights of a free-running step by delay injection: each stage made 2 / 4 us longer (-DLRG_EXP_DELAY_{FRONT,BRANCH,HEAD}=ticks of 10 ns), normal policy
# (results unchanged), steady leg at 68 rooms in flight: d(step time) / d(stage time).
mkdir -p gpurun_out
OUT=gpurun_out/r04_delay.txt
: > $OUT
for V in "base|" "front +2us|-DLRG_EXP_DELAY_FRONT=200" "front +4us|-DLRG_EXP_DELAY_FRONT=400" "branch +2us|-DLRG_EXP_DELAY_BRANCH=200" "branch +4us|-DLRG_EXP_DELAY_BRANCH=400" "head +2us|-DLRG_EXP_DELAY_HEAD=200" "head +4us|-DLRG_EXP_DELAY_HEAD=400" "base|"; do
  NAME="${V%%|*}"; FLAGS="${V#*|}"
  LRG_HIPCC_FLAGS="$FLAGS" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
  for S in ${SLOTS:-68}; do
  LRG_HIPCC_FLAGS="$FLAGS" timeout 600 python bench.py --gpus 1 --rooms $S --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --fixed-rooms 0 > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - <<PY >> $OUT
import json
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print('%-14s slots %3d  %8.0f instance-steps/s  %.2f us/step/slot' % ('$NAME', $S, d['value'], d['us_per_instance_step_per_slot']))
PY
  done
done
LRG_HIPCC_FLAGS="" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cat $OUT
