"""Multi-GPU: rooms are independent units (test_region_grow.py:110-183 re-derives all state per room), so they
shard across ranks with no collective inside the grow loop; the only exchange is the final gather of per-room
labels.  One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU).
The reference has no multi-GPU path at all; this is new design.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_rooms_lpt(sizes, world_size):
    """Longest-processing-time-first assignment of rooms to ranks by equalised point count.
    Returns a list (per rank) of room indices; deterministic, every room assigned exactly once."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    load = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += sizes[i]
    return [sorted(s) for s in shards]


def queue_order(ids, sizes, slots):
    """The order in which a rank's room jobs `ids` (sizes[i] = point count of job i) wait for one of its `slots` (S1, the batched scheduler: rooms are
    independent, test_region_grow.py:110-183, so any order gives the same labels).  Two things decide how long the jobs take together:
    (a) a job must not START so late that it is still running when everything else is done (largest-first, LPT, is the extreme answer);
    (b) a large room's step costs more than a small room's, and not the same workgroups: with the jobs simply sorted by size a launch first holds ONLY large
        rooms and later only small ones -- 0.45 M, then 1.8 M instance-steps/s at 400 slots -- where a mix keeps front workgroups and tile teams both busy
        (1.55 M throughout, profiles/r06_fixed_trace_400*.txt).
    So every job gets a deadline -- the last queue position from which it would still be done with the rest: a job of s points runs for about
    slots x s / sum(sizes) of the whole (steps grow with the point count, and a slot's step takes slots / throughput), and the deadline leaves that much, and a
    quarter more, behind it -- and the positions are filled from the LAST one backwards, each with a job drawn evenly from those whose deadline allows it.  Small
    rooms end up spread over the whole queue, large ones over its front part, and the last positions fall to the smallest by themselves.  Deterministic (a
    golden-ratio sequence, no random state)."""
    desc = sorted(ids, key=lambda i: (-sizes[i], i))
    slots = max(1, int(slots))
    n = len(desc)
    if n <= 2 * slots:
        return desc
    total = float(sum(sizes[i] for i in desc)) or 1.0
    # (the first `slots` positions all start at once: a deadline inside them is as good as position 0)
    deadline = [max(slots - 1, int((1.0 - 1.25 * slots * sizes[i] / total) * n)) for i in desc]      # non-decreasing along desc (sizes fall)
    order = [None] * n
    pool = []                      # jobs that may stand at the position being filled
    nxt = n - 1                    # desc[nxt]: the next job to become eligible (smallest first: the latest deadlines)
    for p in range(n - 1, -1, -1):
        while nxt >= 0 and deadline[nxt] >= p:
            pool.append(desc[nxt])
            nxt -= 1
        if not pool:               # (no job may be this late: the one that minds least)
            pool.append(desc[nxt])
            nxt -= 1
        order[p] = pool.pop(int((((n - p) * 0.6180339887498949) % 1.0) * len(pool)))
    return order


def _short_cut(world, force_collective):
    """A single rank needs no exchange -- unless the caller wants the collectives of an initialised process group executed
    all the same (`force_collective`: the one-rank RCCL run of tests/test_gpu_dist.py and of bench.py at --gpus 1, which is
    the only way the nccl branch can be exercised on a one-GPU box)."""
    return world == 1 and not (force_collective and dist.is_initialized())


def gather_room_labels(local_ids, local_labels, n_rooms, device=None, group=None, force_collective=False):
    """All ranks end up with the labels of all rooms: one all_gather of (room id, size) tables and one
    all_gather of a flat int32 label buffer padded to the largest shard.  `local_labels[i]` is the int array of
    room `local_ids[i]`.  Returns a list of n_rooms arrays (None for rooms nobody owned)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if _short_cut(world, force_collective):
        out = [None] * n_rooms
        for i, lab in zip(local_ids, local_labels):
            out[i] = np.asarray(lab, dtype=np.int32)
        return out
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    meta = torch.full((n_rooms, 2), -1, dtype=torch.int64, device=device)
    for k, (i, lab) in enumerate(zip(local_ids, local_labels)):
        meta[k, 0], meta[k, 1] = i, len(lab)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    totals = [int(m[:, 1].clamp(min=0).sum()) for m in metas]
    flat = torch.zeros(max(max(totals), 1), dtype=torch.int32, device=device)
    if local_labels:
        cat = np.concatenate([np.asarray(l, dtype=np.int32) for l in local_labels]) if totals[dist.get_rank(group)] else np.zeros(0, np.int32)
        flat[:len(cat)] = torch.from_numpy(cat).to(device)
    flats = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(flats, flat, group=group)
    out = [None] * n_rooms
    for m, f in zip(metas, flats):
        m, f = m.cpu().numpy(), f.cpu().numpy()
        o = 0
        for i, n in m:
            if i < 0:
                continue
            out[int(i)] = f[o:o + int(n)].copy()
            o += int(n)
    return out


def gather_flat_labels(local_ids, local_lens, flat, n_rooms, device=None, group=None, force_collective=False):
    """gather_room_labels for labels that are still on the GPU: `flat` is one int32 tensor holding the labels of rooms
    `local_ids` back to back (`local_lens` points each).  With the nccl backend the flat buffer goes into the all_gather as
    it is (device to device over xGMI); gloo stages it through the host.  Returns a list of n_rooms arrays."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if _short_cut(world, force_collective):
        host = flat.cpu().numpy()
        out, o = [None] * n_rooms, 0
        for i, n in zip(local_ids, local_lens):
            out[i] = host[o:o + n]
            o += n
        return out
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    meta = torch.full((n_rooms, 2), -1, dtype=torch.int64)
    if len(local_ids):
        meta[:len(local_ids), 0] = torch.tensor(list(local_ids), dtype=torch.int64)
        meta[:len(local_ids), 1] = torch.tensor(list(local_lens), dtype=torch.int64)
    meta = meta.to(device)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    totals = [int(m[:, 1].clamp(min=0).sum()) for m in metas]
    buf = torch.zeros(max(max(totals), 1), dtype=torch.int32, device=device)
    mine = int(sum(local_lens))
    if mine:
        buf[:mine] = flat[:mine].to(device=device, dtype=torch.int32)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    out = [None] * n_rooms
    for m, f in zip(metas, bufs):
        m, f = m.cpu().numpy(), f.cpu().numpy()
        o = 0
        for i, n in m:
            if i < 0:
                continue
            out[int(i)] = f[o:o + int(n)].copy()
            o += int(n)
    return out


def allreduce_sum(values, device=None, group=None, force_collective=False):
    """Sum a small list of numbers over ranks (throughput accounting)."""
    if not dist.is_initialized() or _short_cut(dist.get_world_size(group), force_collective):
        return list(values)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().tolist()


def allreduce_max(value, device=None, group=None, force_collective=False):
    if not dist.is_initialized() or _short_cut(dist.get_world_size(group), force_collective):
        return value
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
