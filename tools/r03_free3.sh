#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
LRG_FREE_RUN_DEBUG=1 timeout 600 python tools/free_run_perf.py --lockstep 0 --configs 34:3:64,34:1:64,68:1:64 --out gpurun_out/r03_free3_perf.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_free3_perf.log
