#!/bin/bash
# a CU to itself for the front kernel (accounted 128 VGPRs, the default now) and, as a variant, for the medians kernel too: alternating runs
mkdir -p gpurun_out
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration']))"; }
A="--steps 10 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
rm -rf /tmp/exp_a /tmp/exp_b /tmp/exp_c
bash tools/exp_build_run.sh "-DLRG_FRONT_EXCLUSIVE=0" true; cp -r /tmp/exp_repo /tmp/exp_a
bash tools/exp_build_run.sh "-DLRG_FRONT_EXCLUSIVE=1" true; cp -r /tmp/exp_repo /tmp/exp_b
bash tools/exp_build_run.sh "-DLRG_FRONT_EXCLUSIVE=1 -DLRG_BIG_EXCLUSIVE=1" true; cp -r /tmp/exp_repo /tmp/exp_c
v() { d=$1; shift; n=$1; shift; ( cd /tmp/exp_$d && python bench.py $A "$@" 2>/dev/null ) | line "$n"; }
v a "front 88 VGPRs"; v b "front 128"; v c "front 128, medians 128"; v a "front 88 VGPRs"; v b "front 128"; v c "front 128, medians 128"
v b "front 128, 272 rooms" --rooms 272; v a "front 88, 272 rooms" --rooms 272
v b "front 128, scannet" --workload scannet; v a "front 88, scannet" --workload scannet
