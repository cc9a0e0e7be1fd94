#!/bin/bash
# the driver's bench line + a kernel table of the same command (no test suite): gpurun_out/r04_<tag>_bench_default.json, r04_<tag>_bench_kernel_stats.csv
TAG=${1:-x}
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_${TAG}_bench_default.json 2> gpurun_out/bench_default.err || tail -5 gpurun_out/bench_default.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r04_${TAG}_bench_default.json').read().strip().splitlines()[-1])
r = d['roofline']
fw = d['fixed_work']
print('bench: %.0f %s, %.1f us/step/slot; fixed work %d rooms %.0f rooms/s (%.1f waves/rank at 8), best %.0f at %s slots; roofline %.3f; steady192 %s; crc %s; lpt8 %s; cpu %.1f' % (
    d['value'], d['unit'], d['us_per_instance_step_per_slot'], fw['rooms'], fw['rooms_per_sec'], fw['waves_per_rank_at_8_gpus'],
    d.get('fixed_work_best', {}).get('rooms_per_sec', float('nan')), d.get('fixed_work_best', {}).get('slots_per_gpu'), r['frac'],
    d.get('steady_more_rooms_in_flight'), fw['labels_crc32'], fw.get('lpt_balance_at_8_gpus'), d.get('cpu_baseline', {}).get('value', float('nan'))))
PY
if [ "$2" = "prof" ]; then
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_d
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_d -o kt --output-format csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" > /tmp/kt_d.log 2>&1
cp $(ls /tmp/kt_d/*/*kernel_stats.csv /tmp/kt_d/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/r04_${TAG}_bench_kernel_stats.csv
head -6 $R/gpurun_out/r04_${TAG}_bench_kernel_stats.csv | cut -c1-160
fi
