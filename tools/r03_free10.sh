#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 300 python -m pytest tests/test_gpu_free_run.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -3
timeout 500 python tools/free_run_perf.py --jobs 544 --lockstep 1 --configs 34:1:100000:2000,34:1:100000:5000,34:1:100000:20000,68:1:100000:5000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_free10_perf.log
for fd in 6 8; do
  LRG_HIPCC_FLAGS="-DLRG_ASYNC_FD=$fd" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
  echo "== LRG_ASYNC_FD=$fd ==" | tee -a gpurun_out/r03_free10_perf.log
  LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="-DLRG_ASYNC_FD=$fd" timeout 300 python tools/free_run_perf.py --lockstep 0 --seconds 1.0 --configs 34:1:100000:2000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_free10_perf.log
done
