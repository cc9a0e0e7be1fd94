cd learn_region_grow_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w -DLRG_TRACE=1 -o ../liblrg_hip.so lrg_net.hip lrg_fused.hip lrg_grow.hip lrg_grouping.hip; echo rc=$?
cd ../..; python tools/trace_prepare.py 2>&1 | grep -v amdgpu | tail -10
