#!/bin/bash
# Radix-select medians with in-wavefront aggregation of the popular bins: parity tests, Area-5 and KITTI loop rates, kernel tables.
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_grow.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/mf_pytest.log 2>&1
tail -3 gpurun_out/mf_pytest.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/mf_pytest.log | head -10
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration, steady %.1f rooms/s, %s' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration'], d.get('rooms_per_sec_steady_cycling') or 0, d['config']['iteration'][:24]))"; }
timeout 600 python bench.py --steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 2> gpurun_out/mf_a5.err | tee gpurun_out/mf_a5.json | line "area5"
K="--workload kitti --rooms 8 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --policy gt --weights random"
for P in 1 2; do
  timeout 900 python bench.py $K --steps 3 --warmup 1 --packed $P 2> gpurun_out/mf_kitti_$P.err | tee gpurun_out/mf_kitti_packed$P.json | line "kitti packed=$P"
done
cd /tmp && export TMPDIR=/tmp
for P in 1 2; do
  rm -rf /tmp/kt_k$P
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_k$P -o kt --output-format csv -- python $R/bench.py $K --steps 2 --warmup 1 --packed $P > /tmp/kt_k$P.log 2>&1
  cp $(ls /tmp/kt_k$P/*/*kernel_stats.csv /tmp/kt_k$P/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/mf_kitti_packed${P}_kernel_stats.csv
done
rm -rf /tmp/kt_a
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_a -o kt --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 > /tmp/kt_a.log 2>&1
cp $(ls /tmp/kt_a/*/*kernel_stats.csv /tmp/kt_a/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/mf_a5_kernel_stats.csv
cd $R
python - <<'PY'
import csv
for f in ['mf_kitti_packed1', 'mf_kitti_packed2', 'mf_a5']:
    print(f)
    for i, r in enumerate(csv.DictReader(open('gpurun_out/%s_kernel_stats.csv' % f))):
        if i > 7: break
        print('  %-60s %7s %9.1f us  %5s%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
