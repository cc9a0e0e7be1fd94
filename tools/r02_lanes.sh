#!/bin/bash
# bench v2: steady leg by lane count with HIP-graph replays, then one full default line
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for L in 1 2 4; do
  for G in 0 4; do
    timeout 600 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes $L --graph $G > gpurun_out/l_bench_l${L}_g$G.log 2>&1
    echo "lanes $L graph $G: $(tail -1 gpurun_out/l_bench_l${L}_g$G.log | cut -c1-200)"
  done
done
timeout 900 python bench.py > gpurun_out/l_bench_default.log 2>&1; tail -1 gpurun_out/l_bench_default.log | cut -c1-3000
