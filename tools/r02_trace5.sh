#!/bin/bash
# phase stamps of the greedy front kernel: Area-5 set with / without the voxel grid, KITTI scenes
mkdir -p gpurun_out
for V in 0 1; do
  echo "== area5, LRG_NO_VGRID=$V"; LRG_NO_VGRID=$V bash tools/trace_run.sh 1 68 tools/trace_front.py 2>&1 | grep -v amdgpu.ids | tail -14
  rm -rf /tmp/trace_repo
done
echo "== kitti"; LRG_TRACE_WORKLOAD=kitti bash tools/trace_run.sh 1 8 tools/trace_front.py 2>&1 | grep -v amdgpu.ids | tail -14
