#!/bin/bash
# p14_fill under build flags / grid shapes: VARIANTS="name|hipcc flags|ENV=..;..."
mkdir -p gpurun_out
OUT=gpurun_out/r04_fill_ab2.txt
: > $OUT
IFS=';' read -ra VS <<< "$VARIANTS"
for V in "${VS[@]}"; do
  NAME="${V%%|*}"; REST="${V#*|}"; FLAGS="${REST%%|*}"; ENVS="${REST#*|}"
  LRG_HIPCC_FLAGS="$FLAGS" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
  for W in area5 kitti; do
  FR=544; [ $W = kitti ] && FR=64
  env LRG_HIPCC_FLAGS="$FLAGS" $ENVS timeout 900 python bench.py --gpus 1 --workload $W --steps 2 --warmup 1 --best-slots "" --steady-slots "" --cpu-seconds 0 --p0-rooms 0 --fixed-rooms $FR > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - <<PY >> $OUT
import json
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
p = d.get('p14_fill') or {}
print('%-26s %-6s p14 frac %.4f  %.2f ms  crc %s' % ('$NAME', '$W', p.get('frac', 0), 1e3 * p.get('seconds', 0), d['fixed_work']['labels_crc32']))
PY
  done
done
LRG_HIPCC_FLAGS="" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cat $OUT
