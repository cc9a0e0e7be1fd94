#!/bin/bash
# lean packed epilogue: parity tests, bench by lane count, kernel stats at 1 lane, steady-phase pass stamps
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_grow.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -5
for L in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --lanes $L > gpurun_out/pool_bench_l$L.log 2>&1
  echo "lanes $L: $(grep '^{' gpurun_out/pool_bench_l$L.log | tail -1 | cut -c80-330)"
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt_l
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_l -o kt --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes 1 > /tmp/kt_l.log 2>&1
python - <<PY
import csv,glob
f=(glob.glob('/tmp/kt_l/*/*kernel_stats.csv')+glob.glob('/tmp/kt_l/*kernel_stats.csv'))[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('fused_stack','front','gemm')) and int(r['Calls'])>1000:
        print('   1 lane  %-70s calls %6s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
cd $R
