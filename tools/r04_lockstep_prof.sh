#!/bin/bash
# kernel table of the lock-step iterations with many rooms in flight (steady leg of bench.py, --mode lockstep)
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
S=${SLOTS:-544}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_l
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_l -o kt --output-format csv -- python $R/bench.py --gpus 1 --mode lockstep --rooms $S --steps 6 --warmup 2 --cpu-seconds 0 --p0-rooms 0 --best-slots "" --steady-slots "" --fixed-rooms 0 > /tmp/kt_l.log 2>&1
F=$(ls /tmp/kt_l/*/*kernel_stats.csv /tmp/kt_l/*kernel_stats.csv 2>/dev/null | head -1)
head -14 $F | cut -c1-170 > $R/gpurun_out/r04_lockstep_${S}_kernel_stats.csv
cat $R/gpurun_out/r04_lockstep_${S}_kernel_stats.csv
grep '^{' /tmp/kt_l.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['config']['lanes'], d['config']['iterations_per_step'], d['roofline']['frac'])"
