#!/bin/bash
# branch tiles as 1 / 2 / 4 tasks (LrgAsyncBuffers.branch_parts): same results, latency against work
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for p in 0 2 4; do
  echo "== LRG_FREE_RUN_PARTS=$p: tests ==" ; LRG_FREE_RUN_PARTS=$p timeout 400 python -m pytest tests/test_gpu_free_run.py "tests/test_gpu_grow.py::test_packed_iterations_equal_the_nine_launch_step" -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -2
done
export LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="$LRG_HIPCC_FLAGS -DLRG_ASYNC_DEBUG=1"; python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for p in 1 2 4; do
  echo "== kitti, parts $p ==" | tee -a gpurun_out/r03_parts_perf.log
  LRG_FREE_RUN_PARTS=$p timeout 300 python tools/free_run_perf.py --workload kitti --rooms 8 --lockstep 0 --seconds 1.5 --configs 8:1:100000:5000 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03_parts_perf.log
done
for p in 1 2; do
  echo "== area5 68 rooms, parts $p ==" | tee -a gpurun_out/r03_parts_perf.log
  LRG_FREE_RUN_PARTS=$p timeout 300 python tools/free_run_perf.py --lockstep 0 --seconds 1.0 --configs 34:1:100000:5000 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03_parts_perf.log
done
