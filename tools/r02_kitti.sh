#!/bin/bash
# configs[4]: KITTI-shaped scenes (100 k points at 0.3 m), 8 in flight: bench line + kernel table
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1500 python bench.py --workload kitti --rooms 8 --steps 6 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 16 --policy gt --weights random > gpurun_out/r02_bench_kitti_v1.json 2> gpurun_out/kitti.err
tail -1 gpurun_out/r02_bench_kitti_v1.json | cut -c1-500; tail -2 gpurun_out/kitti.err
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt_k
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_k -o kt --output-format csv -- python $R/bench.py --workload kitti --rooms 8 --steps 2 --warmup 1 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --policy gt --weights random > /tmp/kt_k.log 2>&1
cp $(ls /tmp/kt_k/*/*kernel_stats.csv /tmp/kt_k/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/r02_kitti_v1_kernel_stats.csv
head -8 $R/gpurun_out/r02_kitti_v1_kernel_stats.csv | cut -c1-170
