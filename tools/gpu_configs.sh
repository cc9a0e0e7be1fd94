#!/bin/bash
# The other BASELINE.json configurations, one bench line each (on the GPU box): restarts x16, 272 rooms in flight, ScanNet shape.
# (a step is --iters-per-step lock-step iterations; the fixed-work leg is skipped where it does not apply)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
run() { name=$1; shift; timeout 900 python bench.py --cpu-seconds 0 --p0-rooms 0 "$@" > gpurun_out/r02_bench_$name.json 2> gpurun_out/bench_$name.err; tail -1 gpurun_out/r02_bench_$name.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name: %.0f %s, fixed %.0f rooms/s, steady %.0f rooms/s, %.1f us/iteration, slots %s lanes %s' % (d['value'], d['unit'], d.get('rooms_per_sec') or 0, d.get('rooms_per_sec_steady_cycling') or 0, 1e3 * d['ms_per_iteration'], d['config'].get('slots_per_gpu'), d['config'].get('lanes')))"; }
run restart16 --restarts 16 --steps 4 --warmup 2 --iters-per-step 128 --fixed-rooms 0
run 272 --rooms 272 --steps 10 --warmup 4
run scannet --workload scannet --steps 10 --warmup 4
