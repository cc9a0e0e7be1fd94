#!/bin/bash
# Channel-major copy of the centred channels (LrgRoom.chan_major): parity of the loop tests with it, the loop and the KITTI
# configuration with and without it (LRG_NO_CHAN_MAJOR=1 keeps the medians on the [n,F] rows), kernel tables of the KITTI runs.
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_grow.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_beam.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/cm_pytest.log 2>&1
tail -3 gpurun_out/cm_pytest.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/cm_pytest.log | head -10
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration, steady %.1f rooms/s, %s' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration'], d.get('rooms_per_sec_steady_cycling') or 0, d['config']['iteration'][:24]))"; }
for V in 0 1; do
  LRG_NO_CHAN_MAJOR=$V timeout 600 python bench.py --steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 2> gpurun_out/cm_a5_$V.err | tee gpurun_out/cm_a5_nochan$V.json | line "area5 nochan=$V"
done
K="--workload kitti --rooms 8 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 --policy gt --weights random"
for V in 0 1; do
  for P in 1 2; do
    LRG_NO_CHAN_MAJOR=$V timeout 900 python bench.py $K --steps 3 --warmup 1 --packed $P 2> gpurun_out/cm_kitti_$V_$P.err | tee gpurun_out/cm_kitti_nochan${V}_packed$P.json | line "kitti nochan=$V packed=$P"
  done
done
cd /tmp && export TMPDIR=/tmp
for P in 1 2; do
  rm -rf /tmp/kt_k$P
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_k$P -o kt --output-format csv -- python $R/bench.py $K --steps 2 --warmup 1 --packed $P > /tmp/kt_k$P.log 2>&1
  cp $(ls /tmp/kt_k$P/*/*kernel_stats.csv /tmp/kt_k$P/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/cm_kitti_packed${P}_kernel_stats.csv
  echo "packed=$P"; head -9 $R/gpurun_out/cm_kitti_packed${P}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
