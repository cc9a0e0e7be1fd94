#!/bin/bash
# conv[1] hand-over between the branch and the head launch: streaming stores / same problem order (same XCD per tile)
mkdir -p gpurun_out
R=$(pwd)
rm -f $R/gpurun_out/conv1.txt
for V in "0 0" "1 0" "0 1" "1 1"; do
  set -- $V
  rm -rf /tmp/exp_repo; cp -r $R /tmp/exp_repo && cd /tmp/exp_repo
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w -DLRG_NT_CONV1=$1 -DLRG_HEAD_ORDER_AS_BRANCH=$2 -o learn_region_grow_amd/liblrg_hip.so learn_region_grow_amd/csrc/*.hip -Iinclude || exit 1
  export TMPDIR=/tmp
  echo "== nontemporal conv1 stores $1, head problems in branch order $2" | tee -a $R/gpurun_out/conv1.txt
  [ "$V" != "0 0" ] && timeout 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_grow.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -1 | tee -a $R/gpurun_out/conv1.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --lanes 1 --fixed-rooms 0 > /tmp/b.log 2>&1
  echo "   1 lane: $(grep '^{' /tmp/b.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f steps/s  %.1f us/iteration' % (d['value'], 1e3*d['ms_per_iteration']))" 2>&1 | tail -1)" | tee -a $R/gpurun_out/conv1.txt
  rm -rf /tmp/fd_kt
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fd_kt -o kt --output-format csv -- python bench.py --steps 6 --warmup 8 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes 1 > /tmp/fd.log 2>&1
  python - <<PY | tee -a $R/gpurun_out/conv1.txt
import csv,glob
f=(glob.glob('/tmp/fd_kt/*/*kernel_stats.csv')+glob.glob('/tmp/fd_kt/*kernel_stats.csv'))[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('fused_stack','gemm')) and int(r['Calls'])>1000:
        print('   %-70s calls %6s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
  cd $R
done
