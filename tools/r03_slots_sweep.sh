#!/bin/bash
# free-running launches against lock-step iterations by rooms in flight ($1: list of "slots:env" cases separated by ";")
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
IFS=';' read -ra CASES <<< "$1"
for c in "${CASES[@]}"; do
  slots=${c%%:*}; e=${c#*:}; mode=free; [ "$e" = lockstep ] && { mode=lockstep; e=X=1; }
  env $e timeout 900 python bench.py --cpu-seconds 0 --p0-rooms 0 --best-slots= --steady-slots= --rooms $slots --steps 10 --warmup 4 --mode $mode --fixed-rooms 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$slots slots $mode $e: %.0f instance-steps/s' % d['value'])" | tee -a gpurun_out/r03_slots_sweep.log
done
