#!/bin/bash
# The other BASELINE.json configurations, one bench line each (on the GPU box): restarts x16, 272 / 136 rooms in flight (both formulations), ScanNet shape, KITTI shape.
# (a step is a 25 ms free-running launch, or --iters-per-step lock-step iterations; the fixed-work leg is skipped where it does not apply)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
run() { name=$1; shift; timeout 900 python bench.py --cpu-seconds 0 --p0-rooms 0 --best-slots= --steady-slots= "$@" > gpurun_out/r03_bench_$name.json 2> gpurun_out/bench_$name.err; tail -1 gpurun_out/r03_bench_$name.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name: %.0f %s, fixed %.0f rooms/s, steady %.0f rooms/s, %s, slots %s lanes %s' % (d['value'], d['unit'], d.get('rooms_per_sec') or 0, d.get('rooms_per_sec_steady') or d.get('rooms_per_sec_steady_cycling') or 0, d['config']['formulation'][:28], d['config'].get('slots_per_gpu'), d['config'].get('lanes')))" || tail -5 gpurun_out/bench_$name.err; }
run restart16 --restarts 16 --steps 4 --warmup 2 --iters-per-step 128 --fixed-rooms 0
run 272_lockstep --rooms 272 --steps 10 --warmup 4 --mode lockstep --fixed-rooms 0
run 272_free --rooms 272 --steps 10 --warmup 4 --mode free --fixed-rooms 0
run 136_lockstep --rooms 136 --steps 10 --warmup 4 --mode lockstep --fixed-rooms 0
run 136_free --rooms 136 --steps 10 --warmup 4 --mode free --fixed-rooms 0
run scannet --workload scannet --steps 10 --warmup 4
run kitti --workload kitti --rooms 8 --steps 6 --warmup 3 --fixed-rooms 16
