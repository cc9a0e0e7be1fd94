#!/bin/bash
# every counter pass the bench line quotes, on the current build: HBM traffic of the dense evaluation, HBM traffic and SQ counters of the loop
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
bash tools/pmc_run.sh fused gpurun_out/r02_traffic_fused.json 2>&1 | tail -12
bash tools/pmc_traffic_loop.sh gpurun_out/r02_traffic_loop.json 2>&1 | tail -8
bash tools/pmc_sq_loop.sh gpurun_out/r02_pmc_sq_loop.csv 2>&1 | tail -8
ls -la gpurun_out/ | grep -i "pmc\|traffic"
