#!/bin/bash
# Semantic-KITTI-shaped scenes (8 x ~100 k points at 0.3 m, configs[4]) under the formulations of the loop
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
C="--workload kitti --rooms 8 --steps 6 --warmup 3 --fixed-rooms 16 --best-slots= --steady-slots= --cpu-seconds 0 --p0-rooms 0"
for v in "lockstep_chunked:--mode lockstep --packed 1" "lockstep_packed:--mode lockstep --packed 2 --iters-per-step 256" "free_4fronts:--mode free --packed 2" ; do
  name=${v%%:*}; flags=${v#*:}
  timeout 600 python bench.py $C $flags > gpurun_out/r03_kitti_$name.json 2> gpurun_out/r03_kitti_$name.err
  python - $name <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r03_kitti_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print('%-18s %8.0f instance-steps/s, %6.1f scenes/s fixed work (%s)' % (sys.argv[1], d['value'], d['rooms_per_sec'], d['config']['formulation']))
except Exception as e:
    print(sys.argv[1], 'failed', e); print(open('gpurun_out/r03_kitti_%s.err' % sys.argv[1]).read()[-1500:])
PY
done
LRG_FREE_RUN_FRONTS=8 timeout 600 python bench.py $C --mode free --packed 2 > gpurun_out/r03_kitti_free_8fronts.json 2> gpurun_out/r03_kitti_free_8fronts.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03_kitti_free_8fronts.json').read().strip().splitlines()[-1])
print('free_8fronts       %8.0f instance-steps/s, %6.1f scenes/s fixed work' % (d['value'], d['rooms_per_sec']))
PY
python tools/grouping_bench.py 2>/dev/null > gpurun_out/r03_grouping_rates.json; python - <<'PY'
import json
d = json.load(open('gpurun_out/r03_grouping_rates.json'))
for k, v in d.items(): print('%-60s %9.1f us  hbm %.4f  valu %s' % (k, v['gpu_us'], v.get('frac_of_hbm_peak', float('nan')), v.get('frac_of_valu_compare_peak')))
PY
