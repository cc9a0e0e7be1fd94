"""Beam-search region growing (the reference's third local-search driver, test_beam_search.py:143-290) on the same
kernels as the greedy / random-restart loop.

Per seed a queue of at most ``beam_width`` masks; every queue entry spawns ``search_width`` stochastic grow steps; the
children whose mask changed are scored by size (``--scoring np``), the best ``beam_width`` form the next queue; the head of
the queue is committed when its bounding box stalls twice or no child survives.  All children of a level -- of every room in
flight -- are one batch: child ``qid * search_width + search_id`` of a group of ``beam_width * search_width`` slots, its random
stream keyed (seed point, child ordinal, level), so the result does not depend on the order or batching (the reference draws
them one after the other from one stream: that order is the ``rng='legacy'`` definition of the greedy path and is not
offered here).  The queue logic (survivors by size, stall test, commit, next seed, child set-up) runs on the device too
(``lrg_beam_advance``, one workgroup per room): a level is one C call and nothing returns to the host in between.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import LrgBeamGroup
from .grow import RegionGrower, _ptr, _stream_ptr


class BeamSearchGrower(RegionGrower):
    """A room in flight = one group of beam_width * search_width slots; a level of every room in flight = one call of
    lrg_beam_level (queue logic, child set-up, the loop's kernels) with no host decision in between.  The host only watches the
    stats ring for finished rooms (as in greedy growing), fills them in and binds the next room."""
    device_bind = False      # (groups of beam x search slots with a queue of their own: bound on the host, bind_group; reset_room eager)

    def __init__(self, net, rooms_in_flight=16, beam_width=3, search_width=3, seed=0, policy='net', resolution=0.1,
                 cluster_threshold=10):
        self.beam_width, self.search_width = int(beam_width), int(search_width)
        if not (1 <= self.beam_width <= 16 and 1 <= self.search_width and self.beam_width * self.search_width <= 64):
            raise ValueError('beam_width <= 16 and beam_width * search_width <= 64')
        super().__init__(net, rooms_in_flight=rooms_in_flight, restarts=1, group_size=self.beam_width * self.search_width,
                         rng='counter', seed=seed, policy=policy, resolution=resolution, cluster_threshold=cluster_threshold,
                         packed=False)     # the levels run through lrg_beam_level
        self.cluster_threshold = cluster_threshold

    def enqueue_iteration(self):
        raise _lib.LrgHipError('BeamSearchGrower advances level by level (run()); the lock-step iteration of RegionGrower does not apply')

    enqueue = enqueue_graph = grow_loaded = enqueue_iteration

    def load_rooms(self, rooms):
        super().load_rooms(rooms)
        self.d_parent = torch.zeros((self.n_groups, self.beam_width, self.cap), dtype=torch.uint8, device=self.dev)
        self.h_groups = (LrgBeamGroup * self.n_groups)()
        for g in range(self.n_groups):
            B = self.h_groups[g]
            B.parent = self.d_parent.data_ptr() + g * self.beam_width * self.cap
            B.cap = self.cap
            B.room = -1
            B.seed = -1
        self.d_groups = torch.from_numpy(np.frombuffer(bytes(self.h_groups), dtype=np.uint8).copy()).to(self.dev)
        return self

    def bind_group(self, g, r):
        """Room r (or -1) to group g: the next lrg_beam_advance picks its first seed (:154-156)."""
        B = self.h_groups[g]
        B.room, B.seed, B.pending, B.done = int(r), -1, 0, 0
        B.level = B.stuck = B.steps = B.nq = 0
        sz = ctypes.sizeof(LrgBeamGroup)
        buf = np.frombuffer(bytes(B), dtype=np.uint8).copy()
        self.d_groups[g * sz:(g + 1) * sz].copy_(torch.from_numpy(buf))
        self.group_room[g] = r

    def enqueue_level(self):
        flags = self.net.forward_flags | (_lib.LRG_FWD_POOL_ZEROED if self.net.mode == 'fused' else 0)
        rc = self.lib.lrg_beam_level(_ptr(self.d_groups), _ptr(self.d_slots), _ptr(self.d_rooms), self.n_groups, self.beam_width,
                                     self.search_width, self.cap, ctypes.byref(self.params), ctypes.byref(self.net._w),
                                     ctypes.byref(self.step_buffers), flags, _stream_ptr(self.dev))
        _lib.check(rc, 'lrg_beam_level')
        self.iterations += 1
        if self.iterations % self.poll_every == 0:
            self._record_poll()

    def run(self, rooms, fill=True):
        """Grow every room once with beam search; RoomResults in input order."""
        with torch.cuda.device(self.dev):
            self.load_rooms(rooms)
            self.reset_state()
            queue = list(range(self.n_rooms))
            for g in range(self.n_groups):
                self.bind_group(g, queue.pop(0) if queue else -1)
            finished = 0
            while finished < self.n_rooms:
                self.enqueue_level()
                for g in self.poll_done():
                    if fill:
                        self.fill(self.group_room[g])
                    finished += 1
                    self.bind_group(g, queue.pop(0) if queue else -1)
            torch.cuda.current_stream(self.dev).synchronize()
            return self.collect(fill)
