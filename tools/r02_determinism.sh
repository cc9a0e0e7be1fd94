#!/bin/bash
# Is the benchmark configuration deterministic?  Repetitions in one process (tools/determinism_check.py): two lanes with and
# without graph replays, both mask policies, one lane; then the loop tests and the loop rates.
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
run() { echo "== $1"; shift; ( "$@" ) 2>&1 | grep -v "amdgpu.ids" | tail -8; }
run "two lanes, graph replays, ground-truth masks" timeout 600 python tools/determinism_check.py 10 gt 2 4
run "two lanes, graph replays, Bernoulli policy" timeout 600 python tools/determinism_check.py 8 net 2 4
run "three lanes, no graph" timeout 600 python tools/determinism_check.py 6 gt 3 0
run "one lane" timeout 600 python tools/determinism_check.py 4 gt 1 4
run "two lanes, no voxel grid" env LRG_NO_VGRID=1 timeout 600 python tools/determinism_check.py 6 gt 2 4
timeout 1500 python -m pytest tests/test_gpu_grow.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_beam.py tests/test_gpu_cli.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/det_pytest.log 2>&1
tail -3 gpurun_out/det_pytest.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/det_pytest.log | head -10
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_configs.py::test_benchmark_configuration_labels -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -1; done
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration, steady %.1f rooms/s, %s' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration'], d.get('rooms_per_sec_steady_cycling') or 0, d['config']['iteration'][:24]))"; }
A="--steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
K="--workload kitti --rooms 8 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --policy gt --weights random --steps 3 --warmup 1"
timeout 600 python bench.py $A 2> gpurun_out/det_a5.err | tee gpurun_out/det_a5.json | line "area5"
LRG_NO_VGRID=1 timeout 600 python bench.py $A 2> gpurun_out/det_a5n.err | tee gpurun_out/det_a5_nogrid.json | line "area5 no grid"
timeout 900 python bench.py $K --packed 2 2> gpurun_out/det_kitti.err | tee gpurun_out/det_kitti_packed2.json | line "kitti packed=2"
