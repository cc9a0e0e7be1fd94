#!/bin/bash
# HBM traffic of one lrg_forward call from PMC counters: separate --pmc passes (no trace domains), calibrated on a
# 256 MiB copy.  usage (on the GPU box): tools/pmc_run.sh <mode fused|streamed> <out.json>
MODE=${1:-fused}; OUT=${2:-gpurun_out/traffic_$MODE.json}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc && mkdir -p /tmp/pmc
N=10
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc/fetch -o f --output-format csv -- python $R/tools/fwd_only.py 68 $MODE $((N-1)) > /tmp/pmc/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc/write -o w --output-format csv -- python $R/tools/fwd_only.py 68 $MODE $((N-1)) > /tmp/pmc/w.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc/calib_fetch -o cf --output-format csv -- python $R/tools/pmc_calib.py > /tmp/pmc/cf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc/calib_write -o cw --output-format csv -- python $R/tools/pmc_calib.py > /tmp/pmc/cw.log 2>&1
for d in fetch write calib_fetch calib_write; do f=$(find /tmp/pmc/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f /tmp/pmc/$d/ 2>/dev/null; done
python $R/tools/pmc_traffic.py /tmp/pmc $N $R/$OUT && cat $R/$OUT | head -30
