#!/bin/bash
# The other BASELINE.json configurations, one bench line each (on the GPU box): restarts x16, 272 rooms, Bernoulli policy,
# ScanNet shape, KITTI shape.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
run() { name=$1; shift; timeout 900 python bench.py --cpu-seconds 0 --p0-rooms 0 "$@" > gpurun_out/bench_$name.log 2>&1; tail -1 gpurun_out/bench_$name.log | cut -c1-330; }
run restart16 --restarts 16 --steps 300 --warmup 30
run 272 --rooms 272 --steps 600 --warmup 60
run net --policy net --steps 1500 --warmup 100
run scannet --workload scannet --steps 1500 --warmup 100
run kitti --workload kitti --steps 300 --warmup 30
