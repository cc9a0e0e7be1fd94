#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export LRG_FREE_RUN_DEBUG=1
for nf in 272; do
  echo "== $nf rooms in flight, 1088 jobs ==" | tee -a gpurun_out/r03_free13_perf.log
  timeout 500 python tools/free_run_perf.py --jobs 1088 --in-flight $nf --lockstep 0 --configs 68:3:100000:5000,128:3:100000:5000,96:2:100000:5000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_free13_perf.log
done
