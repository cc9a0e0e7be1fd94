#!/usr/bin/env python3
"""Synthetic-data training run: rooms -> staged tuples (learn_region_grow_amd.stage = stage_data.py) -> train_region_grow.py ->
weights (a TensorFlow bundle + an .npz), then a look at the region-grow dynamics those weights give under the reference's
Bernoulli policy.  usage: train_synthetic.py [n_rooms] [epochs] [out_prefix]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import stage, workloads, synthetic, checkpoint  # noqa: E402

n_rooms = int(sys.argv[1]) if len(sys.argv) > 1 else 32
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
out = sys.argv[3] if len(sys.argv) > 3 else 'gpurun_out/lrgnet_synthetic'
os.makedirs(os.path.dirname(out) or '.', exist_ok=True)
t0 = time.time()
targets = [synthetic.AREA5_POINTS[(5 * i) % len(synthetic.AREA5_POINTS)] for i in range(n_rooms)]
targets = [min(t, 14000) for t in targets]                     # staging is a Python loop over the room per step: keep rooms moderate
rooms = [workloads.make_room(t, 20000 + i, i) for i, t in enumerate(targets)]
print('rooms: %d (%.0f s)' % (len(rooms), time.time() - t0))
t0 = time.time()
parts = []
for seed in range(2):                                          # two passes with different seeds / mistake rates (multiseed, :69-76)
    for i, r in enumerate(rooms):
        parts.append(stage.stage_room(r['points'], r['obj_id'], np.random.RandomState(1000 * seed + i)))
data = stage.center_tuples(stage.merge(parts))
staged = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'lrg_staged_synthetic.h5')     # hundreds of MB: not an artefact to keep
stage.save_staged(staged, data)
print('staged %d tuples from %d objects (%.0f s) -> %s' % (len(data['points']), len(data['steps']), time.time() - t0, staged))
import train_region_grow  # noqa: E402
ck = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'lrgnet_synthetic.ckpt')
t0 = time.time()
train_region_grow.main(['--staged', staged, '--ckpt', ck, '--epochs', str(epochs)])
print('training: %.0f s' % (time.time() - t0))
w = checkpoint.load_lrgnet_weights(ck)
np.savez_compressed(out + '_weights.npz', **w)
# ---- dynamics under the reference's policy ----
import torch  # noqa: E402
from learn_region_grow_amd.lrgnet import LrgNetHIP  # noqa: E402
from learn_region_grow_amd.grow import LanedRegionGrower  # noqa: E402
from learn_region_grow_amd import metrics  # noqa: E402
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device='cuda:0').load_weights(w)
test = workloads.area5_rooms(12, seed_base=1000, cache_dir='/tmp/lrg_cache')
for policy in ('net', 'gt'):
    t0 = time.time()
    res = LanedRegionGrower(net, rooms_in_flight=12, rng='counter', seed=0, policy=policy).run(test)
    dt = time.time() - t0
    regs = [r for x in res for r in x.regions]
    lab = [r for r in regs if r['labeled']]
    ms = [metrics.room_metrics(t['obj_id'], x.filled_label) for t, x in zip(test, res)]
    print('policy %s: %d rooms, %.1f regions/room (%.1f labeled), steps/room %.0f, steps/region p50 %d, points/labeled region p50 %d, '
          'mIoU %.2f NMI %.2f, %.2f s' % (policy, len(res), len(regs) / len(res), len(lab) / len(res), np.mean([x.total_steps for x in res]),
                                          np.median([r['steps'] for r in regs]), np.median([r['points'] for r in lab]) if lab else 0,
                                          np.mean([m['iou'] for m in ms]), np.mean([m['nmi'] for m in ms]), dt))
