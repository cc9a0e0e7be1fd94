#!/bin/bash
# KITTI-shaped scenes, 8 in flight: stage breakdown of the free-running launches (debug build) -> gpurun_out/r04_kitti_breakdown.log
mkdir -p gpurun_out
export LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="$LRG_HIPCC_FLAGS -DLRG_ASYNC_DEBUG=1"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 600 python bench.py --workload kitti --rooms 8 --steps 6 --warmup 3 --fixed-rooms 0 --best-slots= --steady-slots= --cpu-seconds 0 --p0-rooms 0 > gpurun_out/kd.json 2> gpurun_out/kd.err
python -c "
import json
d=json.loads(open('gpurun_out/kd.json').read().strip().splitlines()[-1]); print(round(d['value']), d['us_per_instance_step_per_slot'], d['roofline']['rows_evaluated_fraction'], d['roofline']['rows_in_tiles_fraction'])"
grep '^{' gpurun_out/kd.err | tee gpurun_out/r04_kitti_breakdown.log
