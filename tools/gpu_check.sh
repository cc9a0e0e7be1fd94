#!/bin/bash
# The round's acceptance run on a GPU box: build, smoke, GPU test suite, the driver's bench line, a kernel table of the same command.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/gpu_check.sh'
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1800 python -m pytest tests -m gpu -q --maxfail=40 --tb=short -p no:cacheprovider > gpurun_out/r05_d_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/r05_d_pytest_gpu.log
tail -3 gpurun_out/r05_d_pytest_gpu.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/r05_d_pytest_gpu.log | head -10
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_d_bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_d_bench_default.json').read().strip().splitlines()[-1])
r = d['roofline']
print('bench: %.0f %s (%s), %.0f rooms/s fixed work (best %.0f at %s slots), %.1f us per step of a slot; roofline %.3f of fp32 MFMA peak in the loop (%s...), dense %.2f; cpu %.1f steps/s' % (
    d['value'], d['unit'], d['config']['formulation'], d['rooms_per_sec'], d.get('fixed_work_best', {}).get('rooms_per_sec', float('nan')),
    d.get('fixed_work_best', {}).get('slots_per_gpu'), d['us_per_instance_step_per_slot'], r['frac'], r['kernel'][:28], r['dense']['frac'],
    d.get('cpu_baseline', {}).get('value', float('nan'))))
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_d
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_d -o kt --output-format csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --one-room-ks= > /tmp/kt_d.log 2>&1
cp $(ls /tmp/kt_d/*/*kernel_stats.csv /tmp/kt_d/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/r05_d_bench_kernel_stats.csv
head -7 $R/gpurun_out/r05_d_bench_kernel_stats.csv | cut -c1-160
