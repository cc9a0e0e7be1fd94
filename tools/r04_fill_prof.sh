#!/bin/bash
# per-kernel times of the 1-NN fill-in (lrg_nn1_fill_batch) inside bench.py's p14_fill measurement
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_f
LRG_FREE_RUN_FILL=0 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_f -o kt --output-format csv -- python $R/bench.py --gpus 1 --workload ${W:-area5} --steps 2 --warmup 1 --best-slots "" --steady-slots "" --cpu-seconds 0 --p0-rooms 0 --fixed-rooms ${FR:-544} > /tmp/kt_f.log 2>&1
F=$(ls /tmp/kt_f/*/*kernel_stats.csv /tmp/kt_f/*kernel_stats.csv 2>/dev/null | head -1)
grep -i "nn1\|Name" $F | cut -c1-200 > $R/gpurun_out/r04_fill_kernel_stats_${W:-area5}.csv
cat $R/gpurun_out/r04_fill_kernel_stats_${W:-area5}.csv
tail -2 /tmp/kt_f.log | cut -c1-300
