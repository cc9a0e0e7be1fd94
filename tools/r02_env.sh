#!/bin/bash
# runtime knobs that might shorten a dependent launch: kernel arguments in device memory, active waiting
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
rm -f gpurun_out/env_knobs.txt
for E in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_KERNARG_POOL_SIZE=4194304" "ROC_ACTIVE_WAIT_TIMEOUT=100"; do
  env $E timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --lanes 1 --fixed-rooms 0 > /tmp/b.log 2>&1
  echo "$E: $(grep '^{' /tmp/b.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f steps/s  %.1f us/iteration' % (d['value'], 1e3*d['ms_per_iteration']))" 2>&1 | tail -1)" | tee -a gpurun_out/env_knobs.txt
done
