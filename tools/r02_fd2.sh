#!/bin/bash
mkdir -p gpurun_out
R=$(pwd)
rm -f $R/gpurun_out/fd2.txt
IFS=";" read -ra CFGS <<< "${FD_CONFIGS:-4 3 4;8 3 8;12 2 8;16 2 8}"
for V in "${CFGS[@]}"; do
  set -- $V
  FLAGS="-DLRG_PACKED_FD=$1 -DLRG_PACKED_OCC=$2 -DLRG_PACKED_HEAD_FD=$3"
  rm -rf /tmp/exp_repo; cp -r $R /tmp/exp_repo && cd /tmp/exp_repo
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w $FLAGS -o learn_region_grow_amd/liblrg_hip.so learn_region_grow_amd/csrc/*.hip -Iinclude || exit 1
  touch learn_region_grow_amd/liblrg_hip.so
  export TMPDIR=/tmp
  rm -rf /tmp/fd_kt
  timeout 600 python bench.py --steps 3 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 > /tmp/fd_plain.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fd_kt -o kt --output-format csv -- python bench.py --steps 3 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 > /tmp/fd.log 2>&1
  echo "== FD $1 OCC $2 HEADFD $3: $(grep '^{' /tmp/fd_plain.log | tail -1 | cut -c80-140)" | tee -a $R/gpurun_out/fd2.txt
  python - <<PY | tee -a $R/gpurun_out/fd2.txt
import csv,glob
f=(glob.glob('/tmp/fd_kt/*/*kernel_stats.csv')+glob.glob('/tmp/fd_kt/*kernel_stats.csv'))[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('fused_stack','front','gemm')) and int(r['Calls'])>1000:
        print('   %-70s calls %6s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
  cd $R
done
