#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 300 python -m pytest tests/test_gpu_free_run.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -3
LRG_FREE_RUN_DEBUG=1 timeout 600 python tools/free_run_perf.py --lockstep 0 --configs 34:1:64,34:1:100000:2000,34:2:100000:2000,34:3:100000:5000,68:1:100000:2000 --out gpurun_out/r03_free4_perf.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_free4_perf.log
