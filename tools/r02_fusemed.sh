#!/bin/bash
# medians in the branch launch (LRG_FUSE_MEDIANS=1) against the (slot, channel) launch (0), both on the radix-select block median
mkdir -p gpurun_out
R=$(pwd)
rm -f $R/gpurun_out/fusemed.txt
for V in ${FUSE_VARIANTS:-1 0}; do
  rm -rf /tmp/exp_repo; cp -r $R /tmp/exp_repo && cd /tmp/exp_repo
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w -DLRG_FUSE_MEDIANS=$V -o learn_region_grow_amd/liblrg_hip.so learn_region_grow_amd/csrc/*.hip -Iinclude || exit 1
  export TMPDIR=/tmp
  echo "== LRG_FUSE_MEDIANS $V" | tee -a $R/gpurun_out/fusemed.txt
  timeout 900 python -m pytest tests/test_gpu_grow.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3 | tee -a $R/gpurun_out/fusemed.txt
  for L in 1 2; do
    timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --lanes $L > /tmp/b.log 2>&1
    echo "   lanes $L: $(grep '^{' /tmp/b.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f steps/s  fixed %.0f rooms/s  steady %.0f rooms/s  %.1f us/iteration' % (d['value'], d['rooms_per_sec'], d['rooms_per_sec_steady_cycling'], 1e3*d['ms_per_iteration']))" 2>&1 | tail -1)" | tee -a $R/gpurun_out/fusemed.txt
  done
  rm -rf /tmp/fd_kt
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fd_kt -o kt --output-format csv -- python bench.py --steps 6 --warmup 8 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes 1 > /tmp/fd.log 2>&1
  python - <<PY | tee -a $R/gpurun_out/fusemed.txt
import csv,glob
f=(glob.glob('/tmp/fd_kt/*/*kernel_stats.csv')+glob.glob('/tmp/fd_kt/*kernel_stats.csv'))[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('fused_stack','front','gemm')) and int(r['Calls'])>1000:
        print('   %-70s calls %6s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
  cd $R
done
