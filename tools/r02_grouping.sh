#!/bin/bash
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_grouping.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8
timeout 900 python tools/grouping_bench.py > gpurun_out/r02_grouping_rates.json 2> gpurun_out/grouping_bench.err; tail -3 gpurun_out/grouping_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_grouping_rates.json'))
for k,v in d.items(): print('%-62s %9.1f us  %7.1f GB/s  frac %.4f  cpu %s' % (k, v['gpu_us'], v.get('GBps',0), v.get('frac_of_hbm_peak',0), '%.0f us'%v['cpu_reference_us'] if 'cpu_reference_us' in v else '-'))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_grp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_grp -o grp --output-format csv -- python $R/tools/grouping_bench.py > /dev/null 2>&1
cp $(ls /tmp/prof_grp/*/*kernel_stats.csv /tmp/prof_grp/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/r02_grouping_kernel_stats.csv
head -9 $R/gpurun_out/r02_grouping_kernel_stats.csv | cut -c1-150
