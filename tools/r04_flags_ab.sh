#!/bin/bash
# steady leg of bench.py at 68 (and SLOTS) rooms in flight under build flags, normal policy: VARIANTS="name|flags;..."
mkdir -p gpurun_out
OUT=gpurun_out/r04_flags_ab.txt
: > $OUT
IFS=';' read -ra VS <<< "$VARIANTS"
for rep in 1 2; do
for V in "${VS[@]}"; do
  NAME="${V%%|*}"; FLAGS="${V#*|}"
  LRG_HIPCC_FLAGS="$FLAGS" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
  for S in ${SLOTS:-68}; do
  LRG_HIPCC_FLAGS="$FLAGS" timeout 600 python bench.py --gpus 1 --rooms $S --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --fixed-rooms ${FIXED:-0} > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - <<PY >> $OUT
import json
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
fw = d.get('fixed_work') or {}
print('%-16s slots %3d  %8.0f instance-steps/s  %.2f us/step/slot %s' % ('$NAME', $S, d['value'], d['us_per_instance_step_per_slot'], ('crc %s' % fw.get('labels_crc32')) if fw.get('rooms') else ''))
PY
  done
done; done
LRG_HIPCC_FLAGS="" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cat $OUT
