#!/bin/bash
mkdir -p gpurun_out
export LRG_TRACE_WARM=4000
rm -rf /tmp/trace_repo
LRG_TRACE_LAYER=4 bash tools/trace_run.sh 2176 68 tools/trace_loop.py 2>&1 | grep -v amdgpu | tail -26 | tee gpurun_out/trace6_branch.txt
