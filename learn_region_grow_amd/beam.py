"""Beam-search region growing (the reference's third local-search driver, test_beam_search.py:143-290) on the same
kernels as the greedy / random-restart loop.

Per seed a queue of at most ``beam_width`` masks; every queue entry spawns ``search_width`` stochastic grow steps; the
children whose mask changed are scored by size (``--scoring np``), the best ``beam_width`` form the next queue; the head of
the queue is committed when its bounding box stalls twice or no child survives.  All children of a level -- of every room in
flight -- are one batch: child ``qid * search_width + search_id`` of a group of ``beam_width * search_width`` slots, its random
stream keyed (seed point, child ordinal, level), so the result does not depend on the order or batching (the reference draws
them one after the other from one stream: that order is the ``rng='legacy'`` definition of the greedy path and is not
offered here).  The queue logic (sort, stall test, commit) runs on the host between levels from the slots' scan results;
masks never leave the device.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import LRG_ACTIVE, LRG_IDLE, LrgSlot
from .grow import RegionGrower, RoomResult, _ptr, _stream_ptr


class BeamSearchGrower(RegionGrower):
    def __init__(self, net, rooms_in_flight=16, beam_width=3, search_width=3, seed=0, policy='net', resolution=0.1,
                 cluster_threshold=10):
        self.beam_width, self.search_width = int(beam_width), int(search_width)
        super().__init__(net, rooms_in_flight=rooms_in_flight, restarts=1, group_size=self.beam_width * self.search_width,
                         rng='counter', seed=seed, policy=policy, resolution=resolution, cluster_threshold=cluster_threshold,
                         packed=False)     # the levels are driven through the separate entry points
        self.cluster_threshold = cluster_threshold

    def enqueue_iteration(self):
        raise _lib.LrgHipError('BeamSearchGrower advances level by level (run()); the lock-step iteration of RegionGrower does not apply')

    enqueue = enqueue_graph = grow_loaded = enqueue_iteration

    # ---- host-side room state -------------------------------------------------------------------------------------------
    def load_rooms(self, rooms):
        super().load_rooms(rooms)
        self.h_vox = self.d_vox.cpu().numpy()
        self.h_obj = [np.asarray(r['obj_id']).astype(np.int64) for r in rooms]
        self.h_order = [np.asarray(r['order']).astype(np.int64) for r in rooms]
        self.d_parent = torch.zeros((self.n_groups, self.beam_width, self.cap), dtype=torch.uint8, device=self.dev)
        self._slot_dtype = np.dtype(LrgSlot)
        self._slot_view = np.frombuffer(self.h_slots, dtype=self._slot_dtype)          # writable view of the host slot array
        return self

    def _start_room(self, g, r):
        self.group_room[g] = r
        self.state[g] = dict(room=r, visited=np.zeros(self.room_n[r], dtype=bool), cursor=0, regions=[], next_id=1, seed=None)

    def _next_seed(self, st):
        """The next unvisited seed in curvature order (:154-156), or None."""
        order, vis = self.h_order[st['room']], st['visited']
        c = st['cursor']
        while c < len(order) and vis[order[c]]:
            c += 1
        st['cursor'] = c + 1
        if c >= len(order):
            return None
        return int(order[c])

    def _begin_seed(self, g, st, seed):
        o = int(self.room_off[st['room']])
        v = self.h_vox[o + seed].astype(np.int64)
        st.update(seed=seed, level=0, stuck=0, steps=0, seq_mn=v.copy(), seq_mx=v.copy(),
                  Q=[dict(score=0, count=1, mn=v.copy(), mx=v.copy(), parent=-1)])       # parent -1: the seed-only mask (:164-174)

    def _commit(self, g, st):
        """visited / label from the head of the queue (:289-293)."""
        r = st['room']
        o, n = int(self.room_off[r]), self.room_n[r]
        head = st['Q'][0]
        if head['parent'] < 0:
            mask_h = np.zeros(n, dtype=bool)
            mask_h[st['seed']] = True
        else:
            mask_h = self.d_parent[g, head['parent'], :n].cpu().numpy().astype(bool)
        st['visited'] |= mask_h
        sel = torch.from_numpy(mask_h).to(self.dev)
        self.d_visited[o:o + n][sel] = 1                                    # :289
        count = int(mask_h.sum())
        labeled = count > self.cluster_threshold
        if labeled:
            self.d_label[o:o + n][sel] = st['next_id']
            st['next_id'] += 1
        st['regions'].append(dict(seed=st['seed'], steps=st['steps'], points=count, labeled=labeled))
        st['seed'] = None

    # ---- one level of every room in flight ------------------------------------------------------------------------------
    def _level(self):
        G, SW, S = self.G, self.search_width, self.S
        lib, st_ptr, P = self.lib, _stream_ptr(), ctypes.byref(self.params)
        active_groups = []
        for g in range(self.n_groups):
            st = self.state[g]
            if st is None:
                continue
            while True:                                   # commit stalled heads / start seeds until the room has work or is done
                if st['seed'] is None:
                    seed = self._next_seed(st)
                    if seed is None:
                        self._finish_room(g, st)
                        st = self.state[g]
                        if st is None:
                            break
                        continue
                    self._begin_seed(g, st, seed)
                head = st['Q'][0]                          # qid == 0 (:188-198)
                if not np.any(head['mn'] < st['seq_mn']) and not np.any(head['mx'] > st['seq_mx']):
                    if st['stuck'] >= 1:
                        self._commit(g, st)
                        continue
                    st['stuck'] += 1
                else:
                    st['stuck'] = 0
                st['seq_mn'] = np.minimum(st['seq_mn'], head['mn'])
                st['seq_mx'] = np.maximum(st['seq_mx'], head['mx'])
                break
            if st is not None:
                active_groups.append(g)
        if not active_groups:
            return False
        # child slots and their masks, vectorised: the ctypes slot array is edited through a NumPy structured view and the
        # parent masks are scattered into the child slots with three indexed copies
        A = self._slot_view
        A['room'][:] = -1
        A['status'][:] = LRG_IDLE
        seed_slots, seed_points, par_slots, par_src = [], [], [], []
        for g in active_groups:
            st = self.state[g]
            r, nq = st['room'], len(st['Q'])
            lo, hi = g * G, g * G + nq * SW
            V = A[lo:hi]
            V['room'], V['status'], V['seed'], V['step'] = r, LRG_ACTIVE, st['seed'], st['level']
            V['restart'] = np.arange(nq * SW)
            V['steps_total'] = V['stuck'] = V['pad'] = V['nc'] = V['ne'] = V['query'] = V['scan_cnt'] = 0
            V['updated'] = -1
            V['target'] = int(self.h_obj[r][st['seed']])
            V['scan_mn'], V['scan_mx'] = 2147483647, -2147483648
            for qid, q in enumerate(st['Q']):
                W = A[lo + qid * SW:lo + (qid + 1) * SW]
                W['count'], W['mn'], W['mx'] = q['count'], q['mn'], q['mx']
                ids = range(lo + qid * SW, lo + (qid + 1) * SW)
                if q['parent'] < 0:
                    seed_slots.extend(ids)
                    seed_points.extend([st['seed']] * SW)
                else:
                    par_slots.extend(ids)
                    par_src.extend([g * self.beam_width + q['parent']] * SW)
        if seed_slots:
            ss = torch.tensor(seed_slots, device=self.dev)
            self.d_cur[ss] = 0
            self.d_cur[ss, torch.tensor(seed_points, device=self.dev)] = 1
        if par_slots:
            self.d_cur[torch.tensor(par_slots, device=self.dev)] = \
                self.d_parent.view(-1, self.cap)[torch.tensor(par_src, device=self.dev)]
        self.d_slots.copy_(torch.from_numpy(np.frombuffer(self.h_slots, dtype=np.uint8)))
        _lib.check(lib.lrg_box_query(_ptr(self.d_slots), _ptr(self.d_rooms), S, self.cap, P, st_ptr), 'lrg_box_query')
        _lib.check(lib.lrg_median(_ptr(self.d_slots), _ptr(self.d_rooms), S, P, _ptr(self.b_center), st_ptr), 'lrg_median')
        _lib.check(lib.lrg_sample(_ptr(self.d_slots), _ptr(self.d_rooms), S, P, _ptr(self.b_sin), _ptr(self.b_snb), st_ptr), 'lrg_sample')
        _lib.check(lib.lrg_gather_center(_ptr(self.d_slots), _ptr(self.d_rooms), S, P, _ptr(self.b_sin), _ptr(self.b_snb),
                                         _ptr(self.b_center), _ptr(self.b_inl), _ptr(self.b_nbr), _ptr(self.b_gtr), _ptr(self.b_gta),
                                         _ptr(self.b_rows_in), _ptr(self.b_rows_nb), st_ptr), 'lrg_gather_center')
        self.net.forward(self.b_inl, self.b_nbr, self.b_add, self.b_rmv, rows_in=self.b_rows_in, rows_nb=self.b_rows_nb)
        _lib.check(lib.lrg_mask_update(_ptr(self.d_slots), _ptr(self.d_rooms), S, P, _ptr(self.b_inl), _ptr(self.b_nbr),
                                       _ptr(self.b_center), _ptr(self.b_add), _ptr(self.b_rmv), _ptr(self.b_gtr), _ptr(self.b_gta),
                                       None, None, _ptr(self.b_sin), _ptr(self.b_snb), _ptr(self.d_stats), st_ptr), 'lrg_mask_update')
        _lib.check(lib.lrg_bbox_stop(_ptr(self.d_slots), _ptr(self.d_rooms), S, self.cap, P, st_ptr), 'lrg_bbox_stop')
        R = np.frombuffer(self.d_slots.cpu().numpy().tobytes(), dtype=self._slot_dtype)
        upd, cnt = R['updated'].reshape(self.n_groups, G), R['scan_cnt'].reshape(self.n_groups, G)
        src_slots, dst_rows = [], []
        for g in active_groups:
            st = self.state[g]
            nq = len(st['Q'])
            u, c = upd[g, :nq * SW], cnt[g, :nq * SW]
            ran = (u.reshape(nq, SW) >= 0).any(axis=1)             # a parent without neighbours spawns nothing (:212)
            st['steps'] += SW * int(ran.sum())                     # :274, one per child
            alive = np.nonzero((u == 1) & (c > 0))[0]
            if len(alive) == 0:
                self._commit(g, st)                                 # the queue ran dry: its last head is the answer (:179, :289)
                continue
            keep = alive[np.argsort(-c[alive], kind='stable')][:self.beam_width]      # ties keep the (qid, search id) order (:286)
            newQ = []
            for k, ci in enumerate(keep):
                sl = R[g * G + int(ci)]
                newQ.append(dict(score=int(c[ci]), count=int(c[ci]), mn=sl['scan_mn'].astype(np.int64), mx=sl['scan_mx'].astype(np.int64),
                                 parent=k))
                src_slots.append(g * G + int(ci))
                dst_rows.append(g * self.beam_width + k)
            st['Q'] = newQ
            st['level'] += 1
        if src_slots:
            # the survivors become the parents of the next level (gathered out of the child slots before those are reused)
            self.d_parent.view(-1, self.cap)[torch.tensor(dst_rows, device=self.dev)] = self.d_cur[torch.tensor(src_slots, device=self.dev)]
        return True

    def _finish_room(self, g, st):
        r = st['room']
        self.results[r] = st['regions']
        if self._fill:
            self.fill(r)
        self._done += 1
        nxt = self._queue.pop(0) if self._queue else None
        if nxt is None:
            self.state[g] = None
            self.group_room[g] = -1
        else:
            self._start_room(g, nxt)

    def run(self, rooms, fill=True):
        """Grow every room once with beam search; RoomResults in input order."""
        self.load_rooms(rooms)
        self._fill = bool(fill)
        self.state = [None] * self.n_groups
        self.results = [None] * self.n_rooms
        self._queue = list(range(self.n_rooms))
        self._done = 0
        for g in range(self.n_groups):
            if self._queue:
                self._start_room(g, self._queue.pop(0))
        while self._level():
            pass
        torch.cuda.synchronize()
        label = self.d_label.cpu().numpy()
        filled = self.d_filled.cpu().numpy() if fill else None
        out = []
        for r in range(self.n_rooms):
            o, n = int(self.room_off[r]), self.room_n[r]
            out.append(RoomResult(self.room_ids[r], label[o:o + n].astype(np.int64),
                                  filled[o:o + n].astype(np.int64) if fill else None, self.results[r]))
        return out
