#!/bin/bash
# medians in the branch launch: workgroups per slot (LRG_MED_SPLIT) against tile placement and the tiles' wait
mkdir -p gpurun_out
R=$(pwd)
export LRG_TRACE_WARM=4000
rm -f gpurun_out/medsplit.txt
for SP in 9 3; do
  rm -rf /tmp/trace_repo
  echo "== LRG_MED_SPLIT $SP (trace of one steady-phase iteration)" | tee -a gpurun_out/medsplit.txt
  LRG_EXTRA_FLAGS="-DLRG_MED_SPLIT=$SP" LRG_TRACE_LAYER=4 bash tools/trace_run.sh 2176 68 tools/trace_loop.py 2>&1 | grep -v "amdgpu\|tiles with\|pass stamps" | tail -8 | tee -a gpurun_out/medsplit.txt
done
