#!/bin/bash
# pass-level cycle stamps of the packed branch stack (layer 4) and head stack (layer 0 and 1) as the loop launches them
mkdir -p gpurun_out
rm -rf /tmp/trace_repo
LRG_TRACE_LAYER=4 bash tools/trace_run.sh 2176 68 tools/trace_loop.py 2>&1 | tail -34 | tee gpurun_out/trace2_branch.txt
rm -rf /tmp/trace_repo
LRG_TRACE_LAYER=0 bash tools/trace_run.sh 8320 68 tools/trace_loop.py 2>&1 | tail -34 | tee gpurun_out/trace2_head_l0.txt
rm -rf /tmp/trace_repo
LRG_TRACE_LAYER=1 bash tools/trace_run.sh 8320 68 tools/trace_loop.py 2>&1 | tail -34 | tee gpurun_out/trace2_head_l1.txt
