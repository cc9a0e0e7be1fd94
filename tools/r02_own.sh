#!/bin/bash
# medians in the slot's own workgroup (no launch of their own): loop tests, determinism, loop rate against the build with the launch
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1200 python -m pytest tests/test_gpu_grow.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py tests/test_gpu_beam.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/own_pytest.log 2>&1
tail -3 gpurun_out/own_pytest.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/own_pytest.log | head -10
timeout 300 python tools/determinism_check.py 8 gt 2 4 2>&1 | grep -v amdgpu.ids | tail -3
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration, steady %.1f rooms/s' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration'], d.get('rooms_per_sec_steady_cycling') or 0))"; }
A="--steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
for V in 1 0 1 0; do
  bash tools/exp_build_run.sh "-DLRG_FRONT_OWN_MEDIANS=$V" python bench.py $A 2>/dev/null | line "own medians $V"
done
bash tools/exp_build_run.sh "-DLRG_FRONT_OWN_MEDIANS=1" python bench.py $A --lanes 1 2>/dev/null | line "own medians 1, 1 lane"
bash tools/exp_build_run.sh "-DLRG_FRONT_OWN_MEDIANS=0" python bench.py $A --lanes 1 2>/dev/null | line "own medians 0, 1 lane"
