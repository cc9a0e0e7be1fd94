#!/bin/bash
# upper bound of what removing the medians launch from the chain would buy (ground-truth masks: the dynamics do not depend on the centres
# beyond the un-centring quirk)
mkdir -p gpurun_out
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration']))"; }
A="--steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --policy gt --weights random"
for L in 2 1; do
bash tools/exp_build_run.sh "-DLRG_EXP_X=1" python bench.py $A --lanes $L 2> gpurun_out/nb_a$L.err | line "with the medians launch, $L lane(s)"
bash tools/exp_build_run.sh "-DLRG_EXP_NO_BIG_LAUNCH=1" python bench.py $A --lanes $L 2> gpurun_out/nb_b$L.err | line "without it, $L lane(s)"
done
