#!/bin/bash
# where a tile's time goes inside the free-running kernel: cycle stamps of the branch tile (LRG_TRACE=2176) and the head tile (8320)
mkdir -p gpurun_out
export LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="$LRG_HIPCC_FLAGS -DLRG_ASYNC_DEBUG=1"; python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for cap in 2176 8320; do
  LRG_HIPCC_FLAGS="-DLRG_TRACE=$cap" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
  echo "== LRG_TRACE=$cap ==" | tee -a gpurun_out/r03_free7_perf.log
  LRG_HIPCC_FLAGS="-DLRG_TRACE=$cap" timeout 300 python tools/free_run_perf.py --lockstep 0 --seconds 1.0 --configs 34:1:100000:2000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_free7_perf.log
done
