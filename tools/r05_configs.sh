#!/bin/bash
# The other BASELINE.json configurations, one complete bench line each: KITTI shape (8 scenes, speculation by default), ScanNet shape, 16 restarts.  gpurun_out/r05_<name>_bench.json
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
run() { name=$1; shift; timeout 1200 python bench.py --gpus 1 "$@" > gpurun_out/r05_${name}_bench.json 2> gpurun_out/bench_$name.err; tail -1 gpurun_out/r05_${name}_bench.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
o = d.get('one_room_per_gpu') or {}
print('$name: %.0f %s, fixed %.1f rooms/s, roofline %.3f, %s, slots %s, speculation %s; one room: %s' % (d['value'], d['unit'], d.get('rooms_per_sec') or 0, d['roofline']['frac'], d['config']['formulation'][:28], d['config'].get('slots_per_gpu'), d['config'].get('speculation_depth'),
      {k: (v['best_depth'], round(v['speedup_over_one_chain'], 2)) for k, v in o.items() if isinstance(v, dict) and 'best_depth' in v}))" || tail -5 gpurun_out/bench_$name.err; }
run kitti --workload kitti --rooms 8 --steps 12 --warmup 4 --fixed-rooms 64 --p0-rooms 0
run scannet --workload scannet --steps 12 --warmup 4 --p0-rooms 0 --cpu-seconds 0
run restart16 --restarts 16 --steps 4 --warmup 2 --iters-per-step 128 --fixed-rooms 0 --cpu-seconds 0 --p0-rooms 0 --best-slots= --steady-slots=
