#!/bin/bash
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_grow.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py -m gpu -q --tb=short -x -p no:cacheprovider > gpurun_out/f2_pytest.log 2>&1
tail -4 gpurun_out/f2_pytest.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/f2_pytest.log | head -20
for L in 1 2; do
  timeout 600 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes $L --graph 0 > gpurun_out/f2_bench_l${L}.log 2>&1
  echo "lanes $L: $(tail -1 gpurun_out/f2_bench_l${L}.log | cut -c1-200)"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/f2_kt -o kt --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes 1 --graph 0 > $R/gpurun_out/f2_kt.log 2>&1
cd $R
python tools/kt_gaps.py $(ls gpurun_out/f2_kt/*/*kernel_trace.csv gpurun_out/f2_kt/*kernel_trace.csv 2>/dev/null | head -1) 1000 | tee gpurun_out/f2_gaps.txt | head -9
bash tools/trace_run.sh 1 68 tools/trace_front.py 2>&1 | tail -16 | tee gpurun_out/trace_front2.txt
