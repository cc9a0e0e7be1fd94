#!/bin/bash
# bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), here with two ranks on ONE device
# (LRG_BENCH_ONE_DEVICE=1: both ranks use cuda:0, collectives over gloo): the N > 1 code path end to end, and two ranks generating
# the same room set into one cache at the same time
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
rm -rf /tmp/lrg_cache
LRG_BENCH_ONE_DEVICE=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
echo "rc $?"; tail -1 gpurun_out/bench_2rank.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
f = d['fixed_work']
print('2 ranks on one device: %.0f %s, fixed work %d rooms %.0f rooms/s over %d ranks (%s), all labeled %s' % (d['value'], d['unit'], f['rooms'], f['rooms_per_sec'], f['rccl_ranks'], f['collective_backend'], f['all_rooms_labeled_after_gather']))"
tail -3 gpurun_out/bench_2rank.err | cut -c1-300
