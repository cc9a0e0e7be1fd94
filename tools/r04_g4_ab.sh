#!/bin/bash
# G4 (group_point_grad) at the reference harness shape under build flags: VARIANTS="name|flags;..."
mkdir -p gpurun_out
OUT=gpurun_out/r04_g4_ab.txt
: > $OUT
IFS=';' read -ra VS <<< "$VARIANTS"
for V in "${VS[@]}"; do
  NAME="${V%%|*}"; FLAGS="${V#*|}"
  LRG_HIPCC_FLAGS="$FLAGS" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
  for rep in 1 2; do
  LRG_HIPCC_FLAGS="$FLAGS" timeout 600 python tools/grouping_bench.py /tmp/g.json > /dev/null 2> /tmp/g.err || tail -3 /tmp/g.err
  python - <<PY >> $OUT
import json
d = json.load(open('/tmp/g.json'))
e = d['G4 group_point_grad']
print('%-22s G4 %.1f us  %.3f of HBM peak   (G3 %.1f us, G1 %.1f us)' % ('$NAME', e['gpu_us'], e['frac_of_hbm_peak'], d['G3 group_point']['gpu_us'], d['G1 query_ball_point']['gpu_us']))
PY
  done
  LRG_HIPCC_FLAGS="$FLAGS" timeout 600 python -m pytest tests/test_gpu_grouping.py -x -q -k "grad or group" 2>&1 | tail -1 >> $OUT
done
LRG_HIPCC_FLAGS="" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cat $OUT
