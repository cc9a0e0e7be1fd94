#!/bin/bash
# regions up to LRG_FRONT_SMALL points: medians by nine wavefronts of the slot's own workgroup (the launch of the (slot, channel) medians stays
# for the larger ones).  Parity of that path and what it costs / buys at 256.
mkdir -p gpurun_out
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration']))"; }
A="--steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
bash tools/exp_build_run.sh "-DLRG_FRONT_SMALL=256" bash -c "timeout 900 python -m pytest tests/test_gpu_grow.py tests/test_gpu_configs.py -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -6; python bench.py $A 2>/dev/null" | tee gpurun_out/small256.log | tail -8 | line "LRG_FRONT_SMALL=256"
grep -E "passed|failed" gpurun_out/small256.log | tail -2
bash tools/exp_build_run.sh "-DLRG_FRONT_SMALL=0" python bench.py $A 2>/dev/null | line "LRG_FRONT_SMALL=0"
bash tools/exp_build_run.sh "-DLRG_FRONT_SMALL=256" python bench.py $A 2>/dev/null | line "LRG_FRONT_SMALL=256"
bash tools/exp_build_run.sh "-DLRG_FRONT_SMALL=0" python bench.py $A 2>/dev/null | line "LRG_FRONT_SMALL=0"
