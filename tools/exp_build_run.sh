#!/bin/bash
# build a copy of the library with extra -D flags in /tmp and run a command against it:  tools/exp_build_run.sh "-DX=1" python bench.py ...
FLAGS=$1; shift
R=$GRAFT_REPO_ROOT
rm -rf /tmp/exp_repo; cp -r $R /tmp/exp_repo && cd /tmp/exp_repo
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w $FLAGS -o learn_region_grow_amd/liblrg_hip.so learn_region_grow_amd/csrc/*.hip -Iinclude || exit 1
"$@"
