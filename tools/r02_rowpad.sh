#!/bin/bash
# packed rows allocated in multiples of LRG_ROW_PAD: loop tests, then the loop rate at 8 (default build), 16 and 1 (no padding)
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_grow.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_net.py tests/test_gpu_cli.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/rp_pytest.log 2>&1
tail -3 gpurun_out/rp_pytest.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/rp_pytest.log | head -10
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration, steady %.1f rooms/s, packed rows %s' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration'], d.get('rooms_per_sec_steady_cycling') or 0, d['roofline']['in_loop']['packed_rows']))"; }
A="--steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
timeout 600 python bench.py $A 2> gpurun_out/rp_8.err | tee gpurun_out/rp_pad8.json | line "pad 8"
for P in 16 1 8; do
  bash tools/exp_build_run.sh "-DLRG_ROW_PAD=$P" python bench.py $A 2> gpurun_out/rp_$P.err | line "pad $P (exp build)"
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt_a
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_a -o kt --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 > /tmp/kt_a.log 2>&1
cp $(ls /tmp/kt_a/*/*kernel_stats.csv /tmp/kt_a/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/rp_a5_kernel_stats.csv
cd $R
python - <<'PY'
import csv
for i, r in enumerate(csv.DictReader(open('gpurun_out/rp_a5_kernel_stats.csv'))):
    if i > 5: break
    print('  %-60s %7s %9.1f us  %5s%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
