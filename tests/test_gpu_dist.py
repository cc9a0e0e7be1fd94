"""GPU: the RCCL branch of learn_region_grow_amd/dist.py executed on the device.

A one-GPU box cannot hold two RCCL ranks (a communicator refuses two ranks on one device), so the N > 1 exchange is run as
what it is on ONE rank: `init_process_group('nccl', world_size=1, device_id=...)` and the same `all_gather` / `all_reduce`
calls an 8-GPU run makes (`force_collective=True` takes them past the single-rank short cut).  The payload goes device to
device through RCCL's kernels; what a larger world adds is the transport (xGMI), not another code path of this repository."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, socket, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from learn_region_grow_amd import dist as lrg_dist
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%%d' %% port, rank=0, world_size=1, device_id=dev)
sizes = [50, 7, 31, 12, 90]
mine = lrg_dist.shard_rooms_lpt(sizes, 1)[0]
labels = [np.arange(sizes[i], dtype=np.int32) + 1000 * i for i in mine]
flat = torch.from_numpy(np.concatenate(labels)).to(dev)                       # the labels as the growers hold them: on the device
calls = []
orig = dist.all_gather
def counted(out, t, group=None):
    calls.append((t.device.type, t.dtype, tuple(t.shape)))
    return orig(out, t, group=group)
dist.all_gather = counted
got = lrg_dist.gather_flat_labels(mine, [sizes[i] for i in mine], flat, len(sizes), force_collective=True)
got2 = lrg_dist.gather_room_labels(mine, labels, len(sizes), force_collective=True)
dist.all_gather = orig
ok = all(np.array_equal(got[i], np.arange(sizes[i], dtype=np.int32) + 1000 * i) for i in range(len(sizes)))
ok2 = all(np.array_equal(got2[i], np.arange(sizes[i], dtype=np.int32) + 1000 * i) for i in range(len(sizes)))
tot = lrg_dist.allreduce_sum([3.0, 190.0], force_collective=True)
mx = lrg_dist.allreduce_max(2.5, force_collective=True)
# without the switch a single rank takes the short cut (no collective): same result
short = lrg_dist.gather_flat_labels(mine, [sizes[i] for i in mine], flat, len(sizes))
ok3 = all(np.array_equal(short[i], got[i]) for i in range(len(sizes)))
print(json.dumps({'collective_backend': dist.get_backend(), 'ok': bool(ok and ok2 and ok3), 'tot': tot, 'mx': mx, 'n_all_gather': len(calls),
                  'all_on_device': all(c[0] == 'cuda' for c in calls)}))
dist.destroy_process_group()
''' % REPO


@pytest.mark.gpu
def test_label_gather_through_rccl_on_one_rank(cuda_device):
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-c', SCRIPT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert d['collective_backend'] == 'nccl'
    assert d['ok'] and d['tot'] == [3.0, 190.0] and d['mx'] == 2.5
    assert d['n_all_gather'] == 4 and d['all_on_device']          # (room, size) tables + label buffers, twice: every payload a device tensor
