#!/bin/bash
# A/B of build flags on the driver's own line (no --policy gt: real results): VARIANTS="name|flags;..."  ARGS="--rooms 68"   alternating, twice
mkdir -p gpurun_out
OUT=gpurun_out/${OUTNAME:-r05_ab}.txt
: > $OUT
IFS=';' read -ra VS <<< "$VARIANTS"
for rep in 1 2; do
for V in "${VS[@]}"; do
  NAME="${V%%|*}"; FLAGS="${V#*|}"
  LRG_HIPCC_FLAGS="$FLAGS" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
  IFS='@' read -ra AS <<< "${ARGS:---rooms 68}"
  for A in "${AS[@]}"; do
  LRG_HIPCC_FLAGS="$FLAGS" timeout 600 python bench.py --gpus 1 $A --steps 16 --warmup 4 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --one-room-ks= --fixed-rooms ${FIXED:-2176} > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - "$NAME" "$A" <<'PY' >> $OUT
import json, sys
d = json.loads([l for l in open('/tmp/b.json').read().splitlines() if l.startswith('{')][-1])
fw = d.get('fixed_work') or {}
print('%-16s %-34s %9.0f steps/s  %.1f us/step/slot  roofline %.3f | fixed work %.1f rooms/s crc %s' % (sys.argv[1], sys.argv[2], d['value'], d['us_per_instance_step_per_slot'], d['roofline']['frac'], fw.get('rooms_per_sec') or float('nan'), fw.get('labels_crc32')))
PY
  done
done; done
LRG_HIPCC_FLAGS="" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cat $OUT
