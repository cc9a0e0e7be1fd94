#!/bin/bash
# the fixed-work leg's label checksum, repeated: N runs per argument set ("$1", ";"-separated bench.py arguments), environment from the caller
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
N=${N:-6}
IFS=';' read -ra SETS <<< "${1:---rooms 136}"
for a in "${SETS[@]}"; do
  for i in $(seq 1 $N); do
    timeout 600 python bench.py --gpus 1 --mode free --steps 2 --warmup 1 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --one-room-ks= --fixed-rooms 2176 $a 2> /tmp/b.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
fw=d['fixed_work']
print('$a run $i: crc %s rooms/s %.1f labeled %s given_up %s' % (fw['labels_crc32'], fw['rooms_per_sec'], fw['all_rooms_labeled_after_gather'], fw['given_up']))
"
  done
done
