#!/bin/bash
# packed-step quick look: two parity tests, loop throughput by lane count, kernel trace digest
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest "tests/test_gpu_net.py::test_forward_packed_equals_dense_rows" "tests/test_gpu_grow.py::test_feature_size_and_lite_variants" -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -5
for L in 1 2 4; do
  timeout 600 python bench.py --steps 2000 --warmup 500 --cpu-seconds 0 --p0-rooms 0 --lanes $L > gpurun_out/q_bench_l$L.log 2>&1
  tail -1 gpurun_out/q_bench_l$L.log | cut -c1-330
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/q_kt -o kt --output-format csv -- python $R/bench.py --steps 300 --warmup 500 --cpu-seconds 0 --p0-rooms 0 --lanes 1 > $R/gpurun_out/q_kt.log 2>&1
cd $R
python tools/kt_gaps.py $(ls gpurun_out/q_kt/*/*kernel_trace.csv gpurun_out/q_kt/*kernel_trace.csv 2>/dev/null | head -1) 3000 | tee gpurun_out/q_gaps.txt | head -16
