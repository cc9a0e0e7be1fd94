#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export LRG_FREE_RUN_DEBUG=1
for poll in 1 4 16; do
  echo "== poll_sleep $poll, 68 rooms ==" | tee -a gpurun_out/r03_free5_perf.log
  LRG_FREE_RUN_POLL=$poll timeout 300 python tools/free_run_perf.py --lockstep 0 --seconds 1.0 --configs 34:1:100000:2000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_free5_perf.log
done
for cus in 256 48; do
  echo "== 8 rooms, $cus workgroups ==" | tee -a gpurun_out/r03_free5_perf.log
  LRG_FREE_RUN_CUS=$cus timeout 300 python tools/free_run_perf.py --rooms 8 --lockstep 0 --seconds 1.0 --configs 8:1:100000:2000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_free5_perf.log
done
