#!/bin/bash
# cycle stamps: front kernel phases (LRG_TRACE=1) and the packed branch / head stacks (LRG_TRACE=2176 / 8320)
mkdir -p gpurun_out
bash tools/trace_run.sh 1 68 tools/trace_front.py 2>&1 | tail -22 | tee gpurun_out/trace_front.txt
rm -rf /tmp/trace_repo
bash tools/trace_run.sh 2176 68 tools/trace_loop.py 2>&1 | tail -6 | tee gpurun_out/trace_branch_packed.txt
rm -rf /tmp/trace_repo
bash tools/trace_run.sh 8320 68 tools/trace_loop.py 2>&1 | tail -6 | tee gpurun_out/trace_head_packed.txt
