#!/bin/bash
# Dense voxel grid (LrgRoom.vgrid): box query of the greedy front kernel from the grid, voxel lookups as one load; bisection for the
# 48-key medians.  Parity tests (the benchmark-configuration test three times over), Area-5 / KITTI loop rates with and without.
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_grow.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_beam.py tests/test_gpu_cli.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/vg_pytest.log 2>&1
tail -3 gpurun_out/vg_pytest.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/vg_pytest.log | head -10
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_configs.py::test_benchmark_configuration_labels -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -1; done
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration, steady %.1f rooms/s, %s' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration'], d.get('rooms_per_sec_steady_cycling') or 0, d['config']['iteration'][:24]))"; }
A="--steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
K="--workload kitti --rooms 8 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --policy gt --weights random --steps 3 --warmup 1"
timeout 600 python bench.py $A 2> gpurun_out/vg_a5.err | tee gpurun_out/vg_a5.json | line "area5 grid"
LRG_NO_VGRID=1 timeout 600 python bench.py $A 2> gpurun_out/vg_a5n.err | tee gpurun_out/vg_a5_nogrid.json | line "area5 no grid"
timeout 600 python bench.py $A --lanes 1 2> gpurun_out/vg_a5l1.err | tee gpurun_out/vg_a5_lane1.json | line "area5 grid, 1 lane"
for P in 1 2; do
  timeout 900 python bench.py $K --packed $P 2> gpurun_out/vg_kitti_$P.err | tee gpurun_out/vg_kitti_packed$P.json | line "kitti packed=$P"
done
LRG_NO_VGRID=1 timeout 900 python bench.py $K --packed 2 2> gpurun_out/vg_kitti_n.err | tee gpurun_out/vg_kitti_packed2_nogrid.json | line "kitti packed=2 no grid"
bash tools/exp_build_run.sh "-DLRG_MED48_BISECT=0" python bench.py $K --packed 2 2> gpurun_out/vg_kitti_r.err | line "kitti packed=2, radix for 48 keys"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_k /tmp/kt_a
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_k -o kt --output-format csv -- python $R/bench.py $K --packed 2 > /tmp/kt_k.log 2>&1
cp $(ls /tmp/kt_k/*/*kernel_stats.csv /tmp/kt_k/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/vg_kitti_packed2_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_a -o kt --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 > /tmp/kt_a.log 2>&1
cp $(ls /tmp/kt_a/*/*kernel_stats.csv /tmp/kt_a/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/vg_a5_kernel_stats.csv
cd $R
python - <<'PY'
import csv
for f in ['vg_kitti_packed2', 'vg_a5']:
    print(f)
    for i, r in enumerate(csv.DictReader(open('gpurun_out/%s_kernel_stats.csv' % f))):
        if i > 6: break
        print('  %-60s %7s %9.1f us  %5s%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
