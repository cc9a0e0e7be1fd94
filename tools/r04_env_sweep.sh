#!/bin/bash
# free-running steady leg of bench.py under sets of environment variables: VARIANTS="name|A=1 B=2;name2|..." SLOTS="192 272"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
OUT=gpurun_out/${OUTNAME:-r04_env_sweep.txt}
: > $OUT
IFS=';' read -ra VS <<< "$VARIANTS"
for S in ${SLOTS:-192 272}; do
for V in "${VS[@]}"; do
  NAME="${V%%|*}"; ENVS="${V#*|}"
  env $ENVS timeout 600 python bench.py --gpus 1 --mode ${MODE:-free} --rooms $S --steps ${STEPS:-12} --warmup 4 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --fixed-rooms ${FIXED:-0} > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - <<PY >> $OUT
import json
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
fw = d.get('fixed_work') or {}
print('slots %3d %-22s: %8.0f instance-steps/s  %.1f us/step/slot  frac %.3f  %s' % ($S, '$NAME', d['value'], d['us_per_instance_step_per_slot'], d['roofline']['frac'],
      ('fixed work %d rooms %.0f rooms/s crc %s' % (fw['rooms'], fw['rooms_per_sec'], fw.get('labels_crc32'))) if fw.get('rooms') else ''))
PY
done; done
cat $OUT
