#!/bin/bash
# kernel-level durations of the grouping ops (tools/grouping_bench.py under rocprofv3 --kernel-trace --stats) -> gpurun_out/r05_grouping_kernel_stats.csv
mkdir -p gpurun_out; R=$(pwd)
LRG_HIPCC_FLAGS="$FLAGS" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
LRG_HIPCC_FLAGS="$FLAGS" timeout 300 python -m pytest tests/test_gpu_grouping.py -x -q 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_g
LRG_HIPCC_FLAGS="$FLAGS" timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_g -o kt --output-format csv -- python $R/tools/grouping_bench.py /tmp/g.json > /tmp/kt_g.log 2>&1
F=$(ls /tmp/kt_g/*/*kernel_stats.csv /tmp/kt_g/*kernel_stats.csv 2>/dev/null | head -1)
grep -i "lrg_\|Name" $F | cut -c1-170 | head -12 | tee $R/gpurun_out/r05_grouping_kernel_stats.csv
python $R/tools/grouping_bench.py $R/gpurun_out/r05_grouping_rates.json > /dev/null 2>&1; python - <<PY
import json
d = json.load(open("$R/gpurun_out/r05_grouping_rates.json"))
for k, v in d.items():
    print("%-62s %s" % (k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("gpu_us", "frac_of_hbm_peak", "three_launches_us", "fused_us", "fused_frac_of_hbm_peak", "frac_of_valu_compare_peak")}))
PY
