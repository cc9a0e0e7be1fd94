#!/bin/bash
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_net.py::test_forward_packed_equals_dense_rows tests/test_gpu_grow.py tests/test_gpu_fullsize.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
timeout 600 python bench.py --steps 3 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 > /tmp/b.log 2>&1; echo "bench: $(grep '^{' /tmp/b.log | tail -1 | cut -c80-140)"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt_l
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_l -o kt --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 > /tmp/kt_l.log 2>&1
python - <<PY
import csv,glob
f=(glob.glob('/tmp/kt_l/*/*kernel_stats.csv')+glob.glob('/tmp/kt_l/*kernel_stats.csv'))[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('fused_stack','front','gemm')) and int(r['Calls'])>1000:
        print('   %-70s calls %6s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
