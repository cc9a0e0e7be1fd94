#!/bin/bash
# the driver's command line, then the same under rocprofv3 --kernel-trace --stats (summary to gpurun_out/, copied to profiles/ by hand)
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -1 gpurun_out/r02_bench_default.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r02
timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/prof_r02 -o bench --output-format csv -- python $R/bench.py --gpus 1 --steps 4 --warmup 2 --cpu-seconds 0 --p0-rooms 0 > $R/gpurun_out/r02_prof_bench.log 2>&1
cp $(ls /tmp/prof_r02/*/*kernel_stats.csv /tmp/prof_r02/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/r02_bench_kernel_stats.csv
head -12 $R/gpurun_out/r02_bench_kernel_stats.csv | cut -c1-160
