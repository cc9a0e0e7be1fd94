#!/bin/bash
# Semantic-KITTI-shaped scenes (8 x ~100 k points at 0.3 m) with the round's later free-running kernel: rate per formulation, then the stage breakdown (debug build)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
C="--workload kitti --rooms 8 --steps 6 --warmup 3 --fixed-rooms 16 --best-slots= --steady-slots= --cpu-seconds 0 --p0-rooms 0"
run() { # name, env, flags
  env $2 timeout 600 python bench.py $C $3 > gpurun_out/r03_kitti2_$1.json 2> gpurun_out/r03_kitti2_$1.err
  python - $1 <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r03_kitti2_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print('%-22s %8.0f instance-steps/s, %6.1f scenes/s fixed work (%s)' % (sys.argv[1], d['value'], d['rooms_per_sec'], d['config']['formulation'][:40]))
except Exception as e:
    print(sys.argv[1], 'failed', e); print(open('gpurun_out/r03_kitti2_%s.err' % sys.argv[1]).read()[-1500:])
PY
}
run lockstep_packed X=1 "--mode lockstep --packed 2 --iters-per-step 256"
run free_default X=1 "--mode free --packed 2"
run free_8fronts LRG_FREE_RUN_FRONTS=8 "--mode free --packed 2"
run free_8fronts_1team "LRG_FREE_RUN_FRONTS=8 LRG_FREE_RUN_TEAMS=1" "--mode free --packed 2"
run free_8fronts_parts1 "LRG_FREE_RUN_FRONTS=8 LRG_FREE_RUN_PARTS=1" "--mode free --packed 2"
export LRG_FREE_RUN_DEBUG=1 LRG_HIPCC_FLAGS="$LRG_HIPCC_FLAGS -DLRG_ASYNC_DEBUG=1"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
run free_8fronts_debug LRG_FREE_RUN_FRONTS=8 "--mode free --packed 2"
grep '^{' gpurun_out/r03_kitti2_free_8fronts_debug.err | tee gpurun_out/r03_kitti2_breakdown.log
