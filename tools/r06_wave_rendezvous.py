"""Round 6: how often (and at which launch) the start rendezvous of a wave-branch launch -- two kernels resident together -- fails.
   python tools/r06_wave_rendezvous.py [launches] [start_wait_us] [steps] [slots]"""
import sys
import os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from learn_region_grow_amd import synthetic, _lib
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.grow import RegionGrower
from test_gpu_grow import WEIGHT_KW, small_room

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 200
wait_us = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
S = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda:0')
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(synthetic.make_synthetic_weights(**WEIGHT_KW))
rooms = [small_room(400 + i, 2500 + 100 * i, room_id=10 + i) for i in range(S)]
gr = RegionGrower(net, rooms_in_flight=S, rng='counter', seed=123, policy='net', free_run=True, free_run_steps=steps)
gr.load_rooms(rooms)
for g in range(S):
    gr.bind(g, g)
gr.async_buffers.start_wait_us = wait_us
bad = []
for k in range(n_launch):
    gr.enqueue_free_run()
    torch.cuda.synchronize()
    st = gr.d_stats.cpu()
    q = gr.a_queue[:160].cpu()
    if int(st[3]):
        bad.append((k, int(q[112]), int(q[48])))
        gr.d_stats[3] = 0
print('launches %d, failed rendezvous %d: (launch, workgroups arrived, abort reason) %s' % (n_launch, len(bad), bad[:20]))
