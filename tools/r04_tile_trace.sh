#!/bin/bash
# cycle stamps inside a branch tile of the free-running kernel, the passes of layer LRG_TRACE_LAYER stamped one by one (after the MFMAs / after the epilogue)
mkdir -p gpurun_out
: > gpurun_out/r04_tile_trace.log
for L in ${LAYERS:-1 3 4}; do
  F="-DLRG_ASYNC_DEBUG=1 -DLRG_TRACE=${CAP:-2176} -DLRG_TRACE_LAYER=$L"
  LRG_HIPCC_FLAGS="$F" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
  echo "== branch tile, passes of layer $L ==" | tee -a gpurun_out/r04_tile_trace.log
  LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="$F" timeout 300 python tools/free_run_perf.py --lockstep 0 --seconds 1.0 --configs 34:1:100000:2000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_tile_trace.log
done
LRG_HIPCC_FLAGS="" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
