#!/bin/bash
# lane counts for the other configurations on the final build
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration']))"; }
A="--cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
for L in 1 2 3; do python bench.py $A --workload scannet --steps 8 --warmup 4 --lanes $L 2>/dev/null | line "scannet (39 slots), $L lanes"; done
for L in 2 3; do python bench.py $A --rooms 272 --steps 6 --warmup 3 --lanes $L 2>/dev/null | line "272 rooms, $L lanes"; done
for L in 2 3; do python bench.py $A --restarts 16 --steps 3 --warmup 2 --iters-per-step 128 --lanes $L 2>/dev/null | line "restarts x16, $L lanes"; done
for L in 1 2; do python bench.py $A --rooms 32 --steps 8 --warmup 4 --lanes $L 2>/dev/null | line "32 rooms, $L lanes"; done
