#!/bin/bash
# eight KITTI-shaped scenes in flight under sets of environment variables: VARIANTS="name|A=1 B=2;..."
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
OUT=gpurun_out/r04_kitti_env.txt
: > $OUT
IFS=';' read -ra VS <<< "$VARIANTS"
for V in "${VS[@]}"; do
  NAME="${V%%|*}"; ENVS="${V#*|}"
  env $ENVS timeout 600 python bench.py --gpus 1 --workload kitti --rooms 8 --steps 12 --warmup 4 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --fixed-rooms 0 > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - <<PY >> $OUT
import json
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print('%-22s: %8.0f instance-steps/s  %.1f us/step/slot' % ('$NAME', d['value'], d['us_per_instance_step_per_slot']))
PY
done
cat $OUT
