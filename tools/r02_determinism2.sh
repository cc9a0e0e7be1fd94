#!/bin/bash
# run-to-run determinism of the other configurations: restarts (the generic front kernels + the group commit), slots re-bound to
# waiting rooms, ScanNet-shaped rooms, KITTI scenes in both formulations -- each with a third stream keeping the chip busy
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
run() { ( timeout 900 python tools/determinism_check.py "$@" ) 2>&1 | grep -v "amdgpu.ids" | tail -6; }
run 4 net 2 4 --hog 1
run 3 gt 2 4 --restarts 4 --hog 1
run 3 net 2 4 --restarts 4 --in-flight 17 --hog 1
run 4 gt 2 4 --in-flight 17 --hog 1
run 4 net 1 0 --workload scannet --in-flight 39 --hog 1
run 3 gt 2 4 --workload kitti --hog 1
