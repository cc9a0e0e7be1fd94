#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for Q in 4 8; do
for L in 3 4 6 8; do
    GPU_MAX_HW_QUEUES=$Q timeout 600 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes $L --graph 0 > gpurun_out/l2_bench_q${Q}_l${L}.log 2>&1
    echo "hwq $Q lanes $L: $(tail -1 gpurun_out/l2_bench_q${Q}_l${L}.log | cut -c80-140)" | tee -a gpurun_out/lanes_sweep2.txt
done
done
