#!/bin/bash
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
for L in 2 3; do
rm -rf /tmp/kt_l
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_l -o kt --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes $L --graph 0 > /tmp/kt_l.log 2>&1
echo "== lanes $L: $(tail -1 /tmp/kt_l.log | cut -c90-140)" | tee -a $R/gpurun_out/lanes_kt.txt
python $R/tools/kt_gaps.py $(ls /tmp/kt_l/*/*kernel_trace.csv /tmp/kt_l/*kernel_trace.csv 2>/dev/null | head -1) 1000 | grep -E "queue|fused_stack|front|gemm" | tee -a $R/gpurun_out/lanes_kt.txt
done
