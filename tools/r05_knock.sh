#!/bin/bash
# knock-out builds under --policy gt at several slot counts: VARIANTS="name|flags;..."  SLOTS="68 320"
mkdir -p gpurun_out
OUT=gpurun_out/${OUTNAME:-r05_knock}.txt
: > $OUT
IFS=';' read -ra VS <<< "$VARIANTS"
for rep in 1 2; do
for V in "${VS[@]}"; do
  NAME="${V%%|*}"; FLAGS="${V#*|}"
  LRG_HIPCC_FLAGS="$FLAGS" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
  for R in ${SLOTS:-68 320}; do
  LRG_HIPCC_FLAGS="$FLAGS" timeout 600 python bench.py --gpus 1 --rooms $R --policy gt --steps 12 --warmup 4 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --fixed-rooms 0 --one-room-ks= > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - <<PY >> $OUT
import json
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print('%-24s %4d slots %9.0f instance-steps/s  %.1f us/step/slot  frac %.3f  %s' % ('$NAME', $R, d['value'], d['us_per_instance_step_per_slot'], d['roofline']['frac'], d['roofline']['kernel'][:24]))
PY
  done
done; done
LRG_HIPCC_FLAGS="" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cat $OUT
