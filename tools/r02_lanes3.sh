#!/bin/bash
# lanes again, now that a front workgroup has its CU to itself and tiles share a CU two at most; determinism of the final build
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration']))"; }
A="--steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
for L in 2 3 4 2 3 1; do python bench.py $A --lanes $L 2>/dev/null | line "$L lanes"; done
timeout 300 python tools/determinism_check.py 8 net 2 4 --hog 1 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tools/determinism_check.py 6 gt 3 4 2>&1 | grep -v amdgpu.ids | tail -2
