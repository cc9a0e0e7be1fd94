#!/bin/bash
# build a traced copy of the library (LRG_TRACE=2 branch / 1 head) in /tmp and run tools/trace_fused.py against it
K=${1:-2176}; B=${2:-68}; SCRIPT=${3:-tools/trace_fused.py}
R=$GRAFT_REPO_ROOT
cp -r $R /tmp/trace_repo && cd /tmp/trace_repo
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w -DLRG_TRACE=$K -DLRG_TRACE_LAYER=${LRG_TRACE_LAYER:-4} $LRG_EXTRA_FLAGS -o learn_region_grow_amd/liblrg_hip.so learn_region_grow_amd/csrc/*.hip -Iinclude || exit 1
LRG_TRACE_KERNEL=$K python $SCRIPT $B
