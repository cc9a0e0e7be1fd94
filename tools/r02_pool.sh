#!/bin/bash
# median workgroups inside the greedy front launch (-DLRG_MED_POOL_KERNEL=1), second build: role inlined, relaxed polls, small pools.
# Loop tests and determinism with a pool of 16, then the loop rate by pool size against the default build.
mkdir -p gpurun_out
B="-DLRG_MED_POOL_KERNEL=1"
LRG_MED_POOL=16 bash tools/exp_build_run.sh "$B" bash -c "timeout 900 python -m pytest tests/test_gpu_grow.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -5; timeout 300 python tools/determinism_check.py 6 net 2 4 --hog 1 2>&1 | grep -v amdgpu.ids | tail -3" | cut -c1-220
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration, steady %.1f rooms/s' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration'], d.get('rooms_per_sec_steady_cycling') or 0))"; }
A="--steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
for P in 16 0 8 32 16 0; do
  LRG_MED_POOL=$P bash tools/exp_build_run.sh "$B" python bench.py $A 2>/dev/null | line "pool kernel, pool $P"
done
bash tools/exp_build_run.sh "-DLRG_X=1" python bench.py $A 2>/dev/null | line "default build"
LRG_MED_POOL=16 bash tools/exp_build_run.sh "$B" python bench.py $A --lanes 1 2>/dev/null | line "pool 16, 1 lane"
bash tools/exp_build_run.sh "-DLRG_X=1" python bench.py $A --lanes 1 2>/dev/null | line "default build, 1 lane"
