#!/bin/bash
# pooled-product units (LRG_FREE_RUN_UNITS: 0 = on where they fit, -1 = off): parity tests, then the steady rate with and without, then the stage breakdown
mkdir -p gpurun_out
OUT=gpurun_out/r03_units_perf.log; : > $OUT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 600 python -m pytest tests/test_gpu_free_run.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -5 | tee -a $OUT
for u in 0 -1 0 -1; do
  LRG_FREE_RUN_UNITS=$u timeout 300 python tools/free_run_perf.py --lockstep 0 --seconds 1.5 --configs ${CFG:-34:2:100000:5000} 2>&1 | grep '^{' | sed "s/^/units=$u /" | tee -a $OUT
done
export LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="$LRG_HIPCC_FLAGS -DLRG_ASYNC_DEBUG=1"; python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for u in 0 -1; do
  LRG_FREE_RUN_UNITS=$u timeout 300 python tools/free_run_perf.py --lockstep 0 --seconds 1.0 --configs ${CFG:-34:2:100000:5000} 2>&1 | grep '^{' | sed "s/^/debug units=$u /" | tee -a $OUT
done
