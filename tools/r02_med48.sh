#!/bin/bash
# 48-key medians by sampled pivots: the direct median test, loop tests, KITTI loop rate against plain bisection, kernel table
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_grow.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/m48_pytest.log 2>&1
tail -3 gpurun_out/m48_pytest.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/m48_pytest.log | head -10
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration, steady %.1f rooms/s' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration'], d.get('rooms_per_sec_steady_cycling') or 0))"; }
K="--workload kitti --rooms 8 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --policy gt --weights random --steps 3 --warmup 1"
for P in 1 2; do timeout 900 python bench.py $K --packed $P 2> gpurun_out/m48_k$P.err | tee gpurun_out/m48_kitti_packed$P.json | line "kitti packed=$P, sampled pivots"; done
bash tools/exp_build_run.sh "-DLRG_MED48_BISECT=1" python bench.py $K --packed 2 2> gpurun_out/m48_b.err | line "kitti packed=2, plain bisection"
cd /tmp && export TMPDIR=/tmp
for P in 1 2; do
rm -rf /tmp/kt_k
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_k -o kt --output-format csv -- python $R/bench.py $K --packed $P > /tmp/kt_k.log 2>&1
cp $(ls /tmp/kt_k/*/*kernel_stats.csv /tmp/kt_k/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/m48_kitti_packed${P}_kernel_stats.csv
done
cd $R
python - <<'PY'
import csv
for P in (1, 2):
    print('packed', P)
    for i, r in enumerate(csv.DictReader(open('gpurun_out/m48_kitti_packed%d_kernel_stats.csv' % P))):
        if i > 6: break
        print('  %-60s %7s %9.1f us  %5s%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
