#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 300 python -m pytest tests/test_gpu_free_run.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -3
export LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="$LRG_HIPCC_FLAGS -DLRG_ASYNC_DEBUG=1"; python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 400 python tools/free_run_perf.py --lockstep 0 --seconds 1.0 --configs 34:1:100000:2000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_free11_perf.log
