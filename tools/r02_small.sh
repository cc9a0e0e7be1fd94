#!/bin/bash
# (a) medians in the front kernel up to LRG_FRONT_SMALL points (the (slot, channel) launch only for larger regions)
# (b) where two lanes' streams come from (torch's pool / the library's own / the steady leg's reused by the fixed-work leg)
mkdir -p gpurun_out
R=$(pwd)
rm -f $R/gpurun_out/small.txt
for V in 0 1024 4096; do
  rm -rf /tmp/exp_repo; cp -r $R /tmp/exp_repo && cd /tmp/exp_repo
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w -DLRG_FRONT_SMALL=$V -o learn_region_grow_amd/liblrg_hip.so learn_region_grow_amd/csrc/*.hip -Iinclude || exit 1
  export TMPDIR=/tmp
  [ $V -ne 0 ] && timeout 600 python -m pytest tests/test_gpu_grow.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -2 | tee -a $R/gpurun_out/small.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --lanes 1 > /tmp/fd_plain.log 2>&1
  rm -rf /tmp/fd_kt
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fd_kt -o kt --output-format csv -- python bench.py --steps 6 --warmup 8 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --lanes 1 > /tmp/fd.log 2>&1
  echo "== LRG_FRONT_SMALL $V: $(grep '^{' /tmp/fd_plain.log | tail -1 | cut -c80-200)" | tee -a $R/gpurun_out/small.txt
  python - <<PY | tee -a $R/gpurun_out/small.txt
import csv,glob
f=(glob.glob('/tmp/fd_kt/*/*kernel_stats.csv')+glob.glob('/tmp/fd_kt/*kernel_stats.csv'))[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('fused_stack','front','gemm')) and int(r['Calls'])>1000:
        print('   %-70s calls %6s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
  cd $R
done
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for LS in torch own reuse; do
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --lanes 2 --lane-streams $LS > /tmp/b.log 2>&1
  echo "2 lanes, streams $LS: $(grep '^{' /tmp/b.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f steps/s  fixed %.0f rooms/s  steady %.0f rooms/s' % (d['value'], d['rooms_per_sec'], d['rooms_per_sec_steady_cycling']))" 2>&1 | tail -1)" | tee -a gpurun_out/small.txt
done
