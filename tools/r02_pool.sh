#!/bin/bash
# median workgroups inside the greedy front launch: a short smoke first (a hang must not cost the box), loop tests, determinism,
# loop rate with the pool (default) and with the launch of their own (LRG_MED_POOL=0)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 || { echo "smoke failed / timed out"; exit 1; }
timeout 900 python -m pytest tests/test_gpu_grow.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pool_pytest.log 2>&1
tail -3 gpurun_out/pool_pytest.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/pool_pytest.log | head -10
timeout 300 python tools/determinism_check.py 8 gt 2 4 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python tools/determinism_check.py 6 net 2 4 2>&1 | grep -v amdgpu.ids | tail -4
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration, steady %.1f rooms/s' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration'], d.get('rooms_per_sec_steady_cycling') or 0))"; }
A="--steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
for P in 64 0 32 128 64 0; do
  LRG_MED_POOL=$P timeout 600 python bench.py $A 2> gpurun_out/pool_$P.err | line "pool $P"
done
LRG_MED_POOL=64 timeout 600 python bench.py $A --lanes 1 2> gpurun_out/pool_l1.err | line "pool 64, 1 lane"
LRG_MED_POOL=0 timeout 600 python bench.py $A --lanes 1 2> gpurun_out/pool_l1n.err | line "pool 0, 1 lane"
