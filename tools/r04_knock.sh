#!/bin/bash
# knock-out builds under --policy gt: VARIANTS="name|flags;..." ; area5 68 slots and (KITTI=1) eight 100 k-point scenes
mkdir -p gpurun_out
OUT=gpurun_out/r04_knock.txt
: > $OUT
IFS=';' read -ra VS <<< "$VARIANTS"
for rep in 1 2; do
for V in "${VS[@]}"; do
  NAME="${V%%|*}"; FLAGS="${V#*|}"
  LRG_HIPCC_FLAGS="$FLAGS" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
  for W in area5 ${KITTI:+kitti}; do
  R=68; [ $W = kitti ] && R=8
  LRG_HIPCC_FLAGS="$FLAGS" timeout 600 python bench.py --gpus 1 --workload $W --rooms $R --policy gt --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --fixed-rooms 0 > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - <<PY >> $OUT
import json
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print('%-24s %-6s %8.0f instance-steps/s  %.1f us/step/slot' % ('$NAME', '$W', d['value'], d['us_per_instance_step_per_slot']))
PY
  done
done; done
LRG_HIPCC_FLAGS="" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cat $OUT
