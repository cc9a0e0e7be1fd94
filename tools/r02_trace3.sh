#!/bin/bash
# pass-level cycle stamps in the loop's steady phase (4000 iterations in: ~220 tiles per launch, < 1 per CU)
mkdir -p gpurun_out
export LRG_TRACE_WARM=4000
rm -rf /tmp/trace_repo
LRG_TRACE_LAYER=4 bash tools/trace_run.sh 2176 68 tools/trace_loop.py 2>&1 | tail -20 | tee gpurun_out/trace4_branch.txt
rm -rf /tmp/trace_repo
LRG_TRACE_LAYER=1 bash tools/trace_run.sh 2176 68 tools/trace_loop.py 2>&1 | tail -20 | tee gpurun_out/trace4_branch_l1.txt
rm -rf /tmp/trace_repo
LRG_TRACE_LAYER=0 bash tools/trace_run.sh 8320 68 tools/trace_loop.py 2>&1 | tail -20 | tee gpurun_out/trace4_head_l0.txt
rm -rf /tmp/trace_repo
LRG_TRACE_LAYER=1 bash tools/trace_run.sh 8320 68 tools/trace_loop.py 2>&1 | tail -20 | tee gpurun_out/trace4_head_l1.txt
