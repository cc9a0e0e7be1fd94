#!/bin/bash
# Shared tail tiles: steady leg + fixed work by slot count, with / without (LRG_FREE_RUN_TAIL_ROWS=0) and by closing time.   gpurun_out/r05_tail_sweep.txt
#   VARIANTS="name|ENV=.. ENV=..;..."   SLOTS="136 272"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
OUT=gpurun_out/${OUTNAME:-r05_tail_sweep}.txt
: > $OUT
IFS=';' read -ra VS <<< "${VARIANTS:-off|LRG_FREE_RUN_TAIL_ROWS=0;on|X=1}"
for S in ${SLOTS:-136 272}; do
for V in "${VS[@]}"; do
  NAME="${V%%|*}"; ENVS="${V#*|}"
  env $ENVS timeout 900 python bench.py --gpus 1 --mode free --rooms $S --steps 12 --warmup 4 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --one-room-ks= --fixed-rooms ${FIXED:-2176} > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - "$S $NAME" <<'PY' >> $OUT
import json, sys
d = json.loads([l for l in open('/tmp/b.json').read().splitlines() if l.startswith('{')][-1])
fw = d.get('fixed_work') or {}
r = d['roofline']
print('%-34s: %9.0f steps/s  roofline %.3f  rows in tiles %.3f  tiles/eval %.2f | fixed work %s rooms %.1f rooms/s crc %s' % (
    sys.argv[1], d['value'], r['frac'], r['rows_in_tiles_fraction'], r['tiles_run_per_stack'] / max(r['evaluations'], 1), fw.get('rooms'), fw.get('rooms_per_sec', float('nan')), fw.get('labels_crc32')))
PY
done
done
cat $OUT
