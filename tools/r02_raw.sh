#!/bin/bash
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_net.py::test_forward_packed_equals_dense_rows tests/test_gpu_grow.py tests/test_gpu_fullsize.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
for L in 1 2 3; do
    timeout 600 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes $L --graph 0 > gpurun_out/raw_bench_l${L}.log 2>&1
    echo "lanes $L: $(tail -1 gpurun_out/raw_bench_l${L}.log | cut -c80-140)"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_l
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_l -o kt --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes 1 --graph 0 > /tmp/kt_l.log 2>&1
python $R/tools/kt_gaps.py $(ls /tmp/kt_l/*/*kernel_trace.csv /tmp/kt_l/*kernel_trace.csv 2>/dev/null | head -1) 1000 | grep -E "queue|fused_stack|front|gemm"
