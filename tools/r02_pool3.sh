#!/bin/bash
# pool size sweep of the second pool build against the default build, alternating (same box, same call)
mkdir -p gpurun_out
B="-DLRG_MED_POOL_KERNEL=1"
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration']))"; }
A="--steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
rm -rf /tmp/exp_pool /tmp/exp_def
bash tools/exp_build_run.sh "$B" true; cp -r /tmp/exp_repo /tmp/exp_pool
bash tools/exp_build_run.sh "-DLRG_X=1" true; cp -r /tmp/exp_repo /tmp/exp_def
pool() { ( cd /tmp/exp_pool && LRG_MED_POOL=$1 python bench.py $A ${2:-} 2>/dev/null ) | line "pool kernel, pool $1 ${2:-}"; }
def() { ( cd /tmp/exp_def && python bench.py $A ${1:-} 2>/dev/null ) | line "default build ${1:-}"; }
def; pool 0; pool 32; pool 48; pool 64; def; pool 96; pool 128; pool 0; pool 64; def
def "--lanes 1"; pool 64 "--lanes 1"; pool 128 "--lanes 1"; pool 0 "--lanes 1"
