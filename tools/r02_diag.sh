#!/bin/bash
# round-2 baseline diagnostics: loop throughput by lane count, kernel trace with gaps, per-phase cycle stamps of the fused kernels
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for L in 1 2; do
  timeout 600 python bench.py --steps 1000 --warmup 300 --cpu-seconds 0 --p0-rooms 0 --lanes $L > gpurun_out/diag_bench_l$L.log 2>&1
  tail -1 gpurun_out/diag_bench_l$L.log | cut -c1-330
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/diag_kt -o kt --output-format csv -- python $R/bench.py --steps 200 --warmup 300 --cpu-seconds 0 --p0-rooms 0 --lanes 1 > $R/gpurun_out/diag_kt.log 2>&1
cd $R
python tools/kt_gaps.py $(ls gpurun_out/diag_kt/*/*kernel_trace.csv gpurun_out/diag_kt/*kernel_trace.csv 2>/dev/null | head -1) 3000 | tee gpurun_out/diag_gaps.txt
bash tools/trace_run.sh 2176 68 tools/trace_loop.py 2>&1 | tail -8 | tee gpurun_out/diag_trace_branch.txt
rm -rf /tmp/trace_repo
bash tools/trace_run.sh 8320 68 tools/trace_loop.py 2>&1 | tail -8 | tee gpurun_out/diag_trace_head.txt
