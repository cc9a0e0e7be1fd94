#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 600 python tools/free_run_perf.py --out gpurun_out/r03_free2_perf.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_free2_perf.log
