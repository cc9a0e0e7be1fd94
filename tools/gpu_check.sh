#!/bin/bash
# The round's acceptance run on a GPU box: build, smoke, GPU test suite, the driver's bench line, a kernel table of the same command.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/gpu_check.sh'
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/pytest_gpu.log | head -10
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r02_bench_default.json').read().strip().splitlines()[-1])
print('bench: %.0f %s, %.0f rooms/s fixed work, %.0f steady, %.1f us per iteration, %d lanes; roofline %.2f of fp32 MFMA peak (dense); cpu %.1f steps/s' % (
    d['value'], d['unit'], d['rooms_per_sec'], d['rooms_per_sec_steady_cycling'], 1e3 * d['ms_per_iteration'], d['config']['lanes'],
    d['roofline']['frac'], d.get('cpu_baseline', {}).get('value', float('nan'))))
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_d
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_d -o kt --output-format csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 > /tmp/kt_d.log 2>&1
cp $(ls /tmp/kt_d/*/*kernel_stats.csv /tmp/kt_d/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/r02_bench_kernel_stats.csv
head -7 $R/gpurun_out/r02_bench_kernel_stats.csv | cut -c1-160
