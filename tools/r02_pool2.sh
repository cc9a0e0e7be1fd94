#!/bin/bash
# diagnosis of the median-pool experiment, the first loop tests under each variant (queue off everywhere: the front role alone)
mkdir -p gpurun_out
T="tests/test_gpu_grow.py::test_counter_stream_matches_oracle"
run() { echo "== $1"; shift; "$@" 2>&1 | tail -3 | cut -c1-200; }
run "(g) pool role inlined, queue off" env LRG_MED_POOL=0 bash tools/exp_build_run.sh "-DLRG_MED_POOL_KERNEL=1 -DLRG_POOL_INLINE=__forceinline__" timeout 600 python -m pytest $T -m gpu -q --tb=line -p no:cacheprovider
run "(h) pool role a stub, queue off" env LRG_MED_POOL=0 bash tools/exp_build_run.sh "-DLRG_MED_POOL_KERNEL=1 -DLRG_POOL_STUB=1" timeout 600 python -m pytest $T -m gpu -q --tb=line -p no:cacheprovider
run "(i) default build" env LRG_MED_POOL=0 bash tools/exp_build_run.sh "-DLRG_X=1" timeout 600 python -m pytest $T -m gpu -q --tb=line -p no:cacheprovider
