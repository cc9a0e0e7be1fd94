#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 300 python -m pytest tests/test_gpu_free_run.py tests/test_gpu_grow.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -12
timeout 300 python tools/free_run_perf.py --lockstep 0 --seconds 1.0 --configs 34:1:100000:2000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_free12_perf.log
for nf in 136 272; do
  echo "== $nf rooms in flight, 1088 jobs ==" | tee -a gpurun_out/r03_free12_perf.log
  timeout 500 python tools/free_run_perf.py --jobs 1088 --in-flight $nf --lockstep 1 --configs 68:1:100000:5000,68:2:100000:5000,68:3:100000:5000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_free12_perf.log
done
