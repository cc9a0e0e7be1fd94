#!/bin/bash
# workgroups beside a tile of the loop's branch / head launches: accounted 168 (as needed: three per CU), 256 (two), 512 VGPRs (one).
# Alternating runs at 68 rooms in flight, then the configurations with several tiles per CU.
mkdir -p gpurun_out
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.0f %s, %.1f us/iteration' % (d['value'], d['unit'], 1e3 * d['ms_per_iteration']))"; }
A="--cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
rm -rf /tmp/exp_0 /tmp/exp_1 /tmp/exp_2
for L in 0 1 2; do bash tools/exp_build_run.sh "-DLRG_TILE_EXCLUSIVE=$L" true; cp -r /tmp/exp_repo /tmp/exp_$L; done
v() { d=$1; shift; n=$1; shift; ( cd /tmp/exp_$d && python bench.py $A "$@" 2>/dev/null ) | line "$n"; }
for r in 1 2; do for L in 0 1 2; do v $L "68 rooms, level $L" --steps 10 --warmup 5; done; done
for L in 0 1; do v $L "272 rooms, level $L" --rooms 272 --steps 6 --warmup 3; done
for L in 0 1; do v $L "restarts x16, level $L" --restarts 16 --steps 3 --warmup 2 --iters-per-step 128; done
for L in 0 1; do v $L "scannet, level $L" --workload scannet --steps 8 --warmup 4; done
for L in 0 1; do v $L "68 rooms, 1 lane, level $L" --steps 8 --warmup 4 --lanes 1; done
