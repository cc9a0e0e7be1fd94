"""CPU: the N>1 path (room sharding + final label gather) with world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from learn_region_grow_amd import dist as lrg_dist
from learn_region_grow_amd.synthetic import AREA5_POINTS


def test_lpt_sharding_is_a_balanced_partition():
    for world in (1, 2, 4, 8):
        shards = lrg_dist.shard_rooms_lpt(AREA5_POINTS, world)
        assert sorted(i for s in shards for i in s) == list(range(len(AREA5_POINTS)))
        loads = [sum(AREA5_POINTS[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(AREA5_POINTS)            # LPT bound
        assert shards == lrg_dist.shard_rooms_lpt(AREA5_POINTS, world)   # deterministic


def test_queue_order_is_a_deterministic_permutation_that_starts_long_jobs_early():
    """dist.queue_order: every job once; a job's position leaves it the time it needs (as far as the jobs allow); the last positions are the smallest rooms;
    the rooms in flight at any time are a mix of sizes (not the size-sorted queue's)."""
    sizes = [AREA5_POINTS[j % len(AREA5_POINTS)] for j in range(2176)]
    ids = list(range(0, 2176, 2))                                    # (a rank's shard: not all ids, not contiguous)
    total = float(sum(sizes[i] for i in ids))
    for slots in (68, 400):
        order = lrg_dist.queue_order(ids, sizes, slots)
        assert sorted(order) == ids
        assert order == lrg_dist.queue_order(list(reversed(ids)), sizes, slots)
        n = len(order)
        late = [p for p, i in enumerate(order) if p > max(slots - 1, int((1.0 - 1.25 * slots * sizes[i] / total) * n))]
        assert all(sizes[order[p]] <= np.percentile([sizes[i] for i in ids], 35) for p in late)      # only small rooms stand behind their deadline
        assert max(sizes[i] for i in order[-slots // 2:]) <= np.percentile([sizes[i] for i in ids], 50)
        first = [sizes[i] for i in order[:slots]]
        assert max(first) == max(sizes[i] for i in ids) and min(first) < np.percentile([sizes[i] for i in ids], 50)      # the largest AND small ones at the start
    few = lrg_dist.queue_order([3, 1, 2], [5, 9, 7, 1], 68)
    assert few == [1, 2, 3]                                          # fewer jobs than two rounds of slots: largest first


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sizes = [50, 7, 31, 12, 90]
    shards = lrg_dist.shard_rooms_lpt(sizes, world)
    mine = shards[rank]
    labels = [np.arange(sizes[i], dtype=np.int32) + 1000 * i for i in mine]     # stand-in per-room labels
    allrooms = lrg_dist.gather_room_labels(mine, labels, len(sizes))
    ok = all(np.array_equal(allrooms[i], np.arange(sizes[i], dtype=np.int32) + 1000 * i) for i in range(len(sizes)))
    # the same through the flat device-buffer form the CLI and bench.py use (labels of the rank's rooms back to back)
    import torch
    flat = torch.from_numpy(np.concatenate(labels)) if labels else torch.zeros(0, dtype=torch.int32)
    allflat = lrg_dist.gather_flat_labels(mine, [sizes[i] for i in mine], flat, len(sizes))
    ok = ok and all(np.array_equal(allflat[i], np.arange(sizes[i], dtype=np.int32) + 1000 * i) for i in range(len(sizes)))
    tot = lrg_dist.allreduce_sum([len(mine), sum(sizes[i] for i in mine)])
    mx = lrg_dist.allreduce_max(float(rank))
    q.put((rank, ok, tot, mx))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world_size_2_gloo():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, tot, mx in res:
        assert ok and tot == [5.0, 190.0] and mx == 1.0


def test_single_process_gather_is_identity():
    import torch
    out = lrg_dist.gather_room_labels([1, 0], [np.array([5, 6]), np.array([7])], 3)
    assert out[0].tolist() == [7] and out[1].tolist() == [5, 6] and out[2] is None
    out = lrg_dist.gather_flat_labels([1, 0], [2, 1], torch.tensor([5, 6, 7], dtype=torch.int32), 3)
    assert out[0].tolist() == [7] and out[1].tolist() == [5, 6] and out[2] is None
