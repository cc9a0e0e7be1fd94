#!/bin/bash
# bench: steady leg by lane count, plain launches vs HIP-graph replays
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 600 python -m pytest tests/test_gpu_grow.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
for L in 1 2 3 4; do
  for G in 0 4 8; do
    timeout 600 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes $L --graph $G > gpurun_out/l_bench_l${L}_g$G.log 2>&1
    echo "lanes $L graph $G: $(tail -1 gpurun_out/l_bench_l${L}_g$G.log | cut -c80-140)" | tee -a gpurun_out/lanes_sweep.txt
  done
done
