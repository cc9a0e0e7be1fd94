#!/bin/bash
# timeline of the loop with 1 / 2 lanes: per-queue busy share and how many kernels are in flight at once
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
for L in 1 2; do
  rm -rf /tmp/kt_$L
  timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_$L -o kt --output-format csv -- python $R/bench.py --steps 4 --warmup 6 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes $L --graph 0 > /tmp/kt_$L.log 2>&1
  echo "== lanes $L: $(grep '^{' /tmp/kt_$L.log | tail -1 | cut -c80-140)"
  python $R/tools/kt_gaps.py $(ls /tmp/kt_$L/*/*kernel_trace.csv /tmp/kt_$L/*kernel_trace.csv 2>/dev/null | head -1) 4000 | tee $R/gpurun_out/lanes_kt2_l$L.txt
done
