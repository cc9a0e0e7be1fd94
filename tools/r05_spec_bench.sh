#!/bin/bash
# Speculation (RegionGrower(speculate=K)) measured: the default line's one_room_per_gpu entry, the Area-5 set at 68 rooms in flight with K = 0 | 2, and eight
# KITTI-shaped scenes with K = 0 .. 4 (steady leg: kept steps/s; fixed work: scenes/s).   gpurun_out/r05_speculation.txt
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
OUT=gpurun_out/r05_speculation.txt
: > $OUT
line() {   # name, bench args...
  NAME=$1; shift
  timeout 900 python bench.py --gpus 1 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" "$@" > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
  python - "$NAME" <<'PY' >> $OUT
import json, sys
d = json.loads([l for l in open('/tmp/b.json').read().splitlines() if l.startswith('{')][-1])
fw = d.get('fixed_work') or {}
sp = (d['roofline'].get('speculation') or {})
print('%-28s: %9.0f kept steps/s  %6.1f us/step/slot  roofline %.3f  fixed work %s rooms %.1f rooms/s crc %s  voided regions %s steps %s' % (
    sys.argv[1], d['value'], d['us_per_instance_step_per_slot'], d['roofline']['frac'], fw.get('rooms'), fw.get('rooms_per_sec', float('nan')), fw.get('labels_crc32'),
    sp.get('regions_voided'), sp.get('steps_voided')))
o = d.get('one_room_per_gpu')
if o and 'error' not in o:
    for name, v in o.items():
        if name == 'what':
            continue
        print('   one room (%s, %d points): best depth %d, %.2f x one chain; %s; labels equal %s' % (
            name, v['points'], v['best_depth'], v['speedup_over_one_chain'],
            ', '.join('K=%s %.0f ms %.0f steps/s' % (k, 1e3 * x['seconds_per_room'], x['committed_steps_per_sec']) for k, x in v['by_speculation_depth'].items()), v['all_labels_equal']))
elif o:
    print('   one_room_per_gpu error', o)
PY
}
line "area5 68 rooms K=0" --steps 12 --warmup 4 --fixed-rooms 544
line "area5 68 rooms K=2" --steps 12 --warmup 4 --fixed-rooms 544 --speculate 2 --one-room-ks=
line "area5 16 rooms K=0" --rooms 16 --steps 12 --warmup 4 --fixed-rooms 128 --one-room-ks=
line "area5 16 rooms K=3" --rooms 16 --steps 12 --warmup 4 --fixed-rooms 128 --speculate 3 --one-room-ks=
for K in 0 2 3 4; do
  line "kitti 8 scenes K=$K" --workload kitti --rooms 8 --steps 12 --warmup 4 --fixed-rooms 16 --speculate $K $( [ $K != 0 ] && echo "--one-room-ks=" )
done
cat $OUT
