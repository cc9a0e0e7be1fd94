"""RegionGrower -- the region-grow loop of test_region_grow.py / test_random_restart.py, batched.

The reference (test_region_grow.py:175-316) grows ONE region of ONE room per ``sess.run``; here
many rooms are in flight, each contributing one group of slots (one slot for greedy growing,
several for random restarts, test_random_restart.py:169-197), and every lock-step iteration is
one pass of HIP kernels over all slots (include/lrg_hip.h).  All loop state lives on the GPU.

Two sources of randomness:
  rng='counter'  device-side counter stream (Philox keyed by room / seed point / restart / step);
                 no host round trip inside an iteration (``lrg_grow_step``)        -- throughput
  rng='legacy'   the reference's own order: one legacy ``numpy.random.RandomState`` per room,
                 consumed choice / choice / random / random per step (:238-267); sampling
                 positions and Bernoulli masks are decided on the host from the GPU's logits
                 exactly as the reference does (scipy-style softmax, float64 uniforms)    -- parity
"""
import ctypes
import os
import time

import numpy as np
import torch

from . import _lib
from ._lib import (LrgRoom, LrgSlot, LrgGrowParams, LrgStepBuffers, LrgPackedBuffers, LrgAsyncBuffers, LRG_ACTIVE, LRG_DONE, LRG_WAIT, LRG_IDLE,
                   LRG_STATS_WORDS, LRG_DONE_RING, REASON_NAMES)
from .lrgnet import _ptr, _stream_ptr

POLICIES = {'net': 0, 'threshold': 1, 'gt': 2}


def _softmax_conf(logits):
    """scipy.special.softmax(x, axis=-1)[:, 1] in float32 (test_region_grow.py:262-263)."""
    m = logits.max(axis=-1, keepdims=True)
    e = np.exp(logits - m)
    return (e / e.sum(axis=-1, keepdims=True))[..., 1]


class RoomResult:
    def __init__(self, room_id, cluster_label, filled_label, regions):
        self.room_id = room_id
        self.cluster_label = cluster_label      # before fill-in (test_region_grow.py:176,:214)
        self.filled_label = filled_label        # after fill-in (:308-316)
        self.regions = regions                  # dicts: seed, steps, points, reason, labeled, restart

    @property
    def total_steps(self):
        return sum(r['steps'] for r in self.regions)


class RegionGrower:
    device_bind = True       # slots are (re)bound by a kernel (lrg_bind_group); a subclass with slot state of its own binds on the host

    def __init__(self, net, rooms_in_flight=64, restarts=1, group_size=None, rng='counter', seed=0, policy='net',
                 resolution=0.1, cluster_threshold=10, max_region_steps=0, advance_rounds=1, pipeline_depth=4,
                 skip_duplicate_rows=True, poll_every=4, packed=None, graph_iterations=0, scoring='np', free_run=None,
                 free_run_steps=1 << 20, free_run_budget_us=5000, free_run_fronts=0, free_run_teams=0, free_run_fill_cus=None, free_run_units=0,
                 speculate=0, free_run_tail_rows=None, free_run_waves=0, room_order='queue'):
        """packed: True / False / None (= whenever it applies: counter stream, fused network, rooms up to 131072 points):
        one iteration = lrg_grow_step_packed (4 launches, network on the packed distinct rows) instead of lrg_grow_step.
        graph_iterations: > 0 replays that many packed iterations per host call from a HIP graph (lrg_step_graph_*).
        free_run: True / None (= where it applies and is the faster formulation: packed greedy growing, lite 0 / 2, at most 240 slots) /
        False: one host call = ONE launch in which every slot takes up to free_run_steps grow steps at its own pace (lrg_grow_async),
        starting none after free_run_budget_us microseconds (0 = no time limit); same results as the lock-step iterations.
        free_run_fill_cus: > 0: that many CUs are left out of the free-running launches, and the fill-ins of finished rooms
        (test_region_grow.py:308-316) run on them beside the next launch instead of between two launches: the caller runs the grower on
        the first stream of fill_streams(device, free_run_fill_cus) (grower.main_stream), the fill-ins go to the second by themselves.
        None = LRG_FREE_RUN_FILL_CUS, else 0 (fill-ins on the launches' stream).  Measured at 68 rooms in flight: 8 CUs 838-843 k against
        836 k instance-steps/s, 541 against 547 rooms/s fixed work; 6 CUs cannot keep up (728 k), 12 cost more than the fill-ins
        (profiles/r03_units_sweep.log) -- an option, off by default.
        speculate: K > 1 = K slots per room in flight (free-running launches only): the regions of a room's next K unvisited seeds grow side by
        side and are committed in seed order; a region that an earlier commit could have influenced (a committed point inside a box it had
        queried) is dropped and grown again -- identical regions and labels (LrgAsyncBuffers.speculate, csrc/lrg_front.inl).  `rooms_in_flight`
        stays the number of ROOMS in flight; K x that many slots.  For few rooms per GPU (one room or scene per GPU: a single chain of dependent
        steps otherwise)."""
        self.speculate = int(speculate) if speculate and int(speculate) > 1 else 0
        if self.speculate:
            if restarts != 1 or rng != 'counter' or (group_size not in (None, 1)) or free_run is False:
                raise ValueError('speculate needs greedy growing (restarts = 1), the counter stream and free-running launches')
            if self.speculate > 16:
                raise ValueError('speculate: at most 16 slots per room')
            group_size = self.speculate      # the host's slot groups: K slots, one room
            free_run = True
        self.lib = _lib.load()
        self.net = net
        self.dev = net.device
        self.rng = rng
        assert rng in ('counter', 'legacy')
        if group_size is None:
            group_size = restarts if rng == 'counter' else 1
        if rng == 'legacy' and group_size != 1:
            raise ValueError("rng='legacy' replays the reference's sequential stream: group_size must be 1")
        self.G = int(group_size)
        self.n_groups = int(rooms_in_flight)
        self.S = self.n_groups * self.G
        self.advance_rounds = int(advance_rounds)
        self.depth = int(pipeline_depth)
        self.poll_every = max(1, int(poll_every))      # iterations between two read-backs of the device stats block
        self.seed = seed
        p = LrgGrowParams()
        p.resolution = resolution
        p.feature_size = net.feature_size
        p.n_inlier = net.num_inlier_points
        p.n_neighbor = net.num_neighbor_points
        p.cluster_threshold = cluster_threshold
        p.restarts = restarts
        p.group_size = 1 if self.speculate else self.G      # (speculation: greedy slots; their grouping is the launcher's: LrgAsyncBuffers.speculate)
        p.max_region_steps = max_region_steps
        p.rng_seed = seed & 0xFFFFFFFF
        p.policy = POLICIES[policy]
        if scoring not in ('np', 'ml'):
            raise ValueError(scoring)
        if scoring == 'ml' and (rng != 'counter' or packed is False):
            raise ValueError("scoring='ml' is accumulated on the device by the packed iteration (counter stream)")
        p.scoring = 1 if scoring == 'ml' and restarts > 1 else 0
        self.scoring = scoring
        self.params = p
        self.policy = policy
        # evaluate LrgNet only on the distinct leading rows of each stacked set (the rest are copies, :240,:252)
        self.skip_duplicate_rows = bool(skip_duplicate_rows) and rng == 'counter' and net.mode == 'fused'
        self.want_packed = packed
        self.graph_iterations = int(graph_iterations)
        self.packed = False
        self._graph = None
        self.want_free_run = free_run
        self.free_run = False
        self.free_run_steps = int(free_run_steps)
        self.free_run_budget_us = int(free_run_budget_us)
        self.free_run_fronts = int(free_run_fronts)
        self.free_run_teams = int(free_run_teams)
        self.free_run_fill_cus = free_run_fill_cus
        self.free_run_units = int(free_run_units)      # 0 = pooled-product units where they fit, -1 = the tile teams' 128-column blocks
        self.free_run_tail_rows = free_run_tail_rows   # shared tail tiles: rows per side (None = by the slot count, 0 = off)
        self.room_order_mode = room_order              # 'queue': dist.queue_order where rooms wait for slots (more than two rounds of them); 'loaded': as loaded
        self.free_run_waves = int(free_run_waves)      # wave-branch launches (LrgAsyncBuffers.branch_waves): 0 = by the slot count, -1 = off, n = on with n wavefronts per wave-branch CU
        self.debug_hook = None      # tests: called once per active slot per legacy iteration with the step's data
        self._rooms_loaded = False

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def free_run_applies(net, rooms, rooms_in_flight, restarts=1, group_size=None, rng='counter', resolution=0.1, packed=None, free_run=None,
                         skip_duplicate_rows=True, **_):
        """Whether a RegionGrower built with these arguments would grow `rooms` with free-running launches (the decision load_rooms takes
        once it has the rooms on the device), from the host arrays alone -- so that a caller who builds its growers around that answer
        (LanedRegionGrower: one free-running lane, or several lock-step lanes) need not build them twice."""
        if free_run is False or rng != 'counter' or net.mode != 'fused' or not skip_duplicate_rows or packed is False or not rooms:
            return False
        G = int(restarts if group_size is None else group_size)
        ns = [int(len(r['points'])) for r in rooms]
        if G != 1 or int(restarts) != 1 or max(net.num_inlier_points, net.num_neighbor_points) > 512 or getattr(net, 'lite', 0) == 1:
            return False
        if max(ns) > (_lib.LRG_PACKED_MAX_POINTS if packed else _lib.LRG_PACKED_AUTO_POINTS):
            return False
        for r in rooms:      # packed voxel words: every room within 2048 x 2048 x 1024 voxels (lrg_voxelize: rint(x / resolution) in float32)
            if len(r['points']):
                v = np.rint(np.asarray(r['points'], dtype=np.float32)[:, :3] / np.float32(resolution))
                if ((v.max(axis=0) - v.min(axis=0)) > np.array([2047, 2047, 1023])).any():
                    return False
        if free_run:
            return True
        return (int(rooms_in_flight) * G <= _lib.LRG_FREE_RUN_AUTO_SLOTS and max(ns) <= _lib.LRG_FREE_RUN_AUTO_POINTS and
                os.environ.get('LRG_FREE_RUN', '1') != '0')

    def load_rooms(self, rooms):
        """rooms: list of dicts with points [n,F] float32, obj_id [n], order [n] (= argsort(curvatures),
        test_region_grow.py:183) and optional room_id.  Uploads them, voxelises (:175) and builds the
        per-room voxel tables."""
        dev, F, S = self.dev, self.net.feature_size, self.S
        ns = [int(len(r['points'])) for r in rooms]
        # every room starts at a multiple of 16 points in the arenas: the packed iteration reads masks, visited flags and packed
        # voxel words four points per load
        pad = [(n + 15) // 16 * 16 for n in ns]
        offs = np.concatenate([[0], np.cumsum(pad)]).astype(np.int64)
        tot = int(offs[-1])
        self.room_n, self.room_off, self.n_rooms = ns, offs, len(rooms)
        self._fill_ws = None
        pts = np.zeros((tot, F), dtype=np.float32)
        obj = np.zeros(tot, dtype=np.int32)
        order = np.zeros(tot, dtype=np.int32)
        for r, room in enumerate(rooms):
            o, n = int(offs[r]), ns[r]
            p_ = np.asarray(room['points'], dtype=np.float32)
            assert p_.shape == (n, F), (p_.shape, F)
            pts[o:o + n] = p_
            obj[o:o + n] = np.asarray(room['obj_id']).astype(np.int32)
            order[o:o + n] = np.asarray(room['order']).astype(np.int32)
        self.d_points = torch.from_numpy(pts).to(dev)
        # channel-major copy of the centred channels 0, 1, 6 .. F-1 (LrgRoom.chan_major): the medians' keys (:241) of a region
        # -- mostly runs of consecutive indices -- then come from a few dense lines per channel instead of one word per row
        cch = [c for c in range(F) if c < 2 or c >= 6]
        self.d_chan = None
        if tot and len(cch) * tot < 2 ** 31 and os.environ.get('LRG_NO_CHAN_MAJOR') != '1':
            self.d_chan = self.d_points[:, cch].t().contiguous()
        self.d_obj = torch.from_numpy(obj).to(dev)
        self.d_order = torch.from_numpy(order).to(dev)
        self.d_vox = torch.empty((tot, 3), dtype=torch.int32, device=dev)
        self.d_pvox = torch.zeros(tot, dtype=torch.int32, device=dev)
        self.d_visited = torch.zeros(tot, dtype=torch.uint8, device=dev)
        self.d_label = torch.zeros(tot, dtype=torch.int32, device=dev)
        self.d_filled = torch.zeros(tot, dtype=torch.int32, device=dev)
        self.d_rlog = torch.zeros((tot, _lib.LRG_LOG_WORDS), dtype=torch.int32, device=dev)
        caps = [max(16, 1 << int(np.ceil(np.log2(2 * n + 1)))) for n in ns]
        hoffs = np.concatenate([[0], np.cumsum(caps)]).astype(np.int64)
        self.d_hkeys = torch.empty(int(hoffs[-1]), dtype=torch.int64, device=dev)
        self.d_hvals = torch.zeros(int(hoffs[-1]), dtype=torch.int32, device=dev)
        self.d_dup = torch.zeros(1, dtype=torch.int32, device=dev)
        self.d_povf = torch.zeros(1, dtype=torch.int32, device=dev)
        st = _stream_ptr(self.dev)
        _lib.check(self.lib.lrg_voxelize(_ptr(self.d_points), tot, F, ctypes.c_float(self.params.resolution),
                                         _ptr(self.d_vox), st), 'lrg_voxelize')
        # packed voxel words (one per point, relative to the room's minimum voxel) when every room fits 2048 x 2048 x 1024 voxels
        h_vox = self.d_vox.cpu().numpy()
        origins, dims, self.have_pvox = [], [], True
        for r in range(len(rooms)):
            v = h_vox[int(offs[r]):int(offs[r]) + ns[r]]
            lo, hi = (v.min(axis=0), v.max(axis=0)) if ns[r] else (np.zeros(3, np.int64), np.zeros(3, np.int64))
            origins.append([int(x) for x in lo])
            dims.append([int(x) for x in (hi - lo + 1)])
            if ns[r] and ((hi - lo) > np.array([2047, 2047, 1023])).any():
                self.have_pvox = False
        # dense voxel grids (LrgRoom.vgrid): 4 bytes per voxel of a room's bounding box -- a few MB per room; rooms whose box
        # exceeds LRG_VGRID_MAX_CELLS (or a set above LRG_VGRID_TOTAL_CELLS in all) keep the hash table and the room-wide pass
        cells = [dims[r][0] * dims[r][1] * dims[r][2] if ns[r] else 0 for r in range(len(rooms))]
        use_grid = [self.have_pvox and 0 < c <= _lib.LRG_VGRID_MAX_CELLS and os.environ.get('LRG_NO_VGRID') != '1' for c in cells]
        if sum(c for c, u in zip(cells, use_grid) if u) > _lib.LRG_VGRID_TOTAL_CELLS:
            use_grid = [False] * len(rooms)
        goffs = np.concatenate([[0], np.cumsum([(c + 15) // 16 * 16 if u else 0 for c, u in zip(cells, use_grid)])]).astype(np.int64)
        self.d_vgrid = torch.empty(int(goffs[-1]), dtype=torch.int32, device=dev) if goffs[-1] else None
        self.h_rooms = (LrgRoom * len(rooms))()
        for r, room in enumerate(rooms):
            o, n, ho = int(offs[r]), ns[r], int(hoffs[r])
            R = self.h_rooms[r]
            if self.have_pvox:
                R.pvox = self.d_pvox.data_ptr() + o * 4
                R.vox_origin[0], R.vox_origin[1], R.vox_origin[2] = origins[r]
                _lib.check(self.lib.lrg_voxel_pack(ctypes.c_void_p(self.d_vox.data_ptr() + o * 12), n, origins[r][0], origins[r][1],
                                                   origins[r][2], ctypes.c_void_p(R.pvox), _ptr(self.d_povf), st), 'lrg_voxel_pack')
            if use_grid[r]:
                R.vgrid = self.d_vgrid.data_ptr() + int(goffs[r]) * 4
                R.vgrid_dim[0], R.vgrid_dim[1], R.vgrid_dim[2] = dims[r]
                _lib.check(self.lib.lrg_voxel_grid_build(ctypes.c_void_p(self.d_vox.data_ptr() + o * 12), n, origins[r][0], origins[r][1],
                                                         origins[r][2], dims[r][0], dims[r][1], dims[r][2], ctypes.c_void_p(R.vgrid), st),
                           'lrg_voxel_grid_build')
            R.points = self.d_points.data_ptr() + o * F * 4
            if self.d_chan is not None:
                R.chan_major = self.d_chan.data_ptr() + o * 4
                R.chan_stride = tot
            R.voxels = self.d_vox.data_ptr() + o * 12
            R.obj_id = self.d_obj.data_ptr() + o * 4
            R.order = self.d_order.data_ptr() + o * 4
            R.visited = self.d_visited.data_ptr() + o
            R.label = self.d_label.data_ptr() + o * 4
            R.hash_keys = self.d_hkeys.data_ptr() + ho * 8
            R.hash_vals = self.d_hvals.data_ptr() + ho * 4
            R.region_log = self.d_rlog.data_ptr() + o * 4 * _lib.LRG_LOG_WORDS
            R.n = n
            R.hash_mask = caps[r] - 1
            R.next_cluster_id = 1
            R.seed_cursor = 0
            R.n_regions = 0
            R.done = 0
            R.room_id = int(room.get('room_id', r))
            _lib.check(self.lib.lrg_voxel_hash_build(ctypes.c_void_p(R.voxels), n, ctypes.c_void_p(R.hash_keys),
                                                     ctypes.c_void_p(R.hash_vals), R.hash_mask, _ptr(self.d_dup), st),
                       'lrg_voxel_hash_build')
        dup = int(self.d_dup.item())
        if dup:
            raise _lib.LrgHipError('room voxels are not unique / out of range (flag %d): rooms must be equalised at '
                                   'the grow resolution (test_region_grow.py:125-134)' % dup)
        self.room_ids = [int(self.h_rooms[r].room_id) for r in range(len(rooms))]
        self.d_rooms = torch.from_numpy(np.frombuffer(bytes(self.h_rooms), dtype=np.uint8).copy()).to(dev)
        # ---- slots ----
        cap = (max(ns) + 15) // 16 * 16           # (lrg_grow_step_packed sets mask bytes with word-wide atomics)
        self.cap = cap
        self.d_cur = torch.zeros((S, cap), dtype=torch.uint8, device=dev)
        self.d_best = torch.zeros((S, cap), dtype=torch.uint8, device=dev)
        self.d_curidx = torch.zeros((S, cap), dtype=torch.int32, device=dev)
        self.d_candidx = torch.zeros((S, cap), dtype=torch.int32, device=dev)
        nchunk = (cap + _lib.LRG_SCAN_CHUNK - 1) // _lib.LRG_SCAN_CHUNK
        self.d_chunkcnt = torch.zeros((S, 2 * nchunk), dtype=torch.int32, device=dev)
        self.h_slots = (LrgSlot * S)()
        for s in range(S):
            sl = self.h_slots[s]
            sl.cur = self.d_cur.data_ptr() + s * cap
            sl.best = self.d_best.data_ptr() + s * cap
            sl.cur_idx = self.d_curidx.data_ptr() + s * cap * 4
            sl.cand_idx = self.d_candidx.data_ptr() + s * cap * 4
            sl.chunk_cnt = self.d_chunkcnt.data_ptr() + s * 2 * nchunk * 4
            sl.room = -1
            sl.status = LRG_IDLE
            sl.seed = -1
        self.d_slots = torch.from_numpy(np.frombuffer(bytes(self.h_slots), dtype=np.uint8).copy()).to(dev)
        # ---- step buffers ----
        Ni, Nn = self.net.num_inlier_points, self.net.num_neighbor_points
        self.b_center = torch.zeros((S, 16), dtype=torch.float32, device=dev)
        self.b_sin = torch.zeros((S, Ni), dtype=torch.int32, device=dev)
        self.b_snb = torch.zeros((S, Nn), dtype=torch.int32, device=dev)
        self.b_inl = torch.zeros((S, Ni, F), dtype=torch.float32, device=dev)
        self.b_nbr = torch.zeros((S, Nn, F), dtype=torch.float32, device=dev)
        self.b_gtr = torch.zeros((S, Ni), dtype=torch.int32, device=dev)
        self.b_gta = torch.zeros((S, Nn), dtype=torch.int32, device=dev)
        self.b_add = torch.zeros((S, Nn, 2), dtype=torch.float32, device=dev)
        self.b_rmv = torch.zeros((S, Ni, 2), dtype=torch.float32, device=dev)
        self.b_amask = torch.zeros((S, Nn), dtype=torch.uint8, device=dev)
        self.b_rmask = torch.zeros((S, Ni), dtype=torch.uint8, device=dev)
        self.d_stats = torch.zeros(LRG_STATS_WORDS, dtype=torch.int64, device=dev)
        self.b_rows_in = torch.zeros(S, dtype=torch.int32, device=dev)
        self.b_rows_nb = torch.zeros(S, dtype=torch.int32, device=dev)
        # a workspace of its own, zero-filled once and only ever used through lrg_grow_step (LRG_FWD_POOL_ZEROED)
        nbytes = self.lib.lrg_forward_workspace_bytes(ctypes.byref(self.net._w), S, Ni, Nn)
        ws = self.d_ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        sb = LrgStepBuffers()
        sb.center, sb.sample_in, sb.sample_nb = self.b_center.data_ptr(), self.b_sin.data_ptr(), self.b_snb.data_ptr()
        sb.inlier, sb.neighbor = self.b_inl.data_ptr(), self.b_nbr.data_ptr()
        sb.gt_remove, sb.gt_add = self.b_gtr.data_ptr(), self.b_gta.data_ptr()
        sb.add_logits, sb.rmv_logits = self.b_add.data_ptr(), self.b_rmv.data_ptr()
        sb.workspace, sb.workspace_bytes = ws.data_ptr(), ws.numel()
        sb.stats = self.d_stats.data_ptr()
        if self.skip_duplicate_rows:
            sb.rows_in, sb.rows_nb = self.b_rows_in.data_ptr(), self.b_rows_nb.data_ptr()
        self.step_buffers = sb
        self._release_graph()
        can_pack = (self.rng == 'counter' and self.net.mode == 'fused' and max(ns) <= _lib.LRG_PACKED_MAX_POINTS and
                    max(Ni, Nn) <= 1024 and self.skip_duplicate_rows and
                    self.lib.lrg_forward_packed_workspace_bytes(ctypes.byref(self.net._w), S, 32) > 0)
        self.free_run = False
        if self.want_free_run and not can_pack:
            raise ValueError('free-running launches need packed iterations')
        if self.want_packed and not can_pack:
            raise ValueError('packed iterations need the counter stream, the fused network and rooms of at most %d points'
                             % _lib.LRG_PACKED_MAX_POINTS)
        self.packed = (can_pack and max(ns) <= _lib.LRG_PACKED_AUTO_POINTS) if self.want_packed is None else bool(self.want_packed)
        if self.params.scoring == 1 and not self.packed:
            raise ValueError("scoring='ml' needs the packed iteration (fused network, rooms of at most %d points)" % _lib.LRG_PACKED_MAX_POINTS)
        if self.packed:
            cap_rows = S * ((max(Ni, Nn) + 31) // 32 * 32)     # (a slot's rows are allocated in multiples of 8 -- or, free-running, have
                                                                 #  a place of their own of whole 32-row tiles)
            # Shared tail tiles of the free-running launches (LrgAsyncBuffers.tail_ctl): rows behind the slots' own that the slots' tails share, so that the rows
            # beyond a slot's last full tile fill tiles together.  Sized for a 25 ms launch at the rate the slot count sustains (~32 k evaluations x 2 x 16 rows:
            # when a launch runs out of them, the slots pad tiles of their own again).  On from 224 slots, where the tile teams are what the launch is bound by and
            # the CUs they save can serve slots instead (44 front workgroups instead of 34: lrg_grow_async): 2 176 room jobs at 272 slots 828 -> 872 rooms/s, 320: ->
            # 886; below, a step is a chain of latencies and a tile that waits for a second slot's tail only lengthens it (68 slots: 868 -> 714 k instance-steps/s,
            # 136: 1.14 -> 1.07 M; profiles/r05_tail_sweep_v2.txt, r05_tail_fronts*.txt).  LRG_FREE_RUN_TAIL_ROWS: 0 = off, n = that many rows per side.
            self.tail_rows = 0
            if self.want_free_run is not False and F >= 9 and F <= 16:
                env = os.environ.get('LRG_FREE_RUN_TAIL_ROWS', '')
                want = int(env) if env else (self.free_run_tail_rows if self.free_run_tail_rows is not None else (min(1 << 20, 4096 * S) if (S >= 224 and not self.speculate) else 0))
                self.tail_rows = max(0, want) // 32 * 32
            cap_rows += self.tail_rows + (32 if self.tail_rows else 0)      # (+ a tile: a head tile on a slot's tail rows stages 32 rows from the tail's first)
            self.row_cap = cap_rows
            # (room for rows at a 64-byte stride: the free-running kernel gathers and stages its rows in 16-byte pieces, LrgAsyncBuffers.rows16; the
            #  lock-step launches use the first cap_rows x F floats of the same arrays)
            self.p_xin = torch.zeros((cap_rows, max(F, 16)), dtype=torch.float32, device=dev)
            self.p_xnb = torch.zeros((cap_rows, max(F, 16)), dtype=torch.float32, device=dev)
            self.p_rsin = torch.zeros(cap_rows, dtype=torch.int32, device=dev)
            self.p_rsnb = torch.zeros(cap_rows, dtype=torch.int32, device=dev)
            self.p_updin = torch.zeros((S, Ni, 4), dtype=torch.float32, device=dev)
            self.p_updnb = torch.zeros((S, Nn, 4), dtype=torch.float32, device=dev)
            self.p_rmv = torch.zeros((cap_rows, 2), dtype=torch.float32, device=dev)
            self.p_add = torch.zeros((cap_rows, 2), dtype=torch.float32, device=dev)
            self.p_slot_rows = torch.zeros((S, 4), dtype=torch.int32, device=dev)
            self.p_counters = torch.zeros(4, dtype=torch.int32, device=dev)
            pbytes = self.lib.lrg_forward_packed_workspace_bytes(ctypes.byref(self.net._w), S, cap_rows)
            self.p_ws = torch.zeros(pbytes, dtype=torch.uint8, device=dev)
            pb = LrgPackedBuffers()
            pb.center, pb.sample_in, pb.sample_nb = self.b_center.data_ptr(), self.b_sin.data_ptr(), self.b_snb.data_ptr()
            pb.x_in, pb.x_nb = self.p_xin.data_ptr(), self.p_xnb.data_ptr()
            pb.row_slot_in, pb.row_slot_nb = self.p_rsin.data_ptr(), self.p_rsnb.data_ptr()
            pb.upd_in, pb.upd_nb = self.p_updin.data_ptr(), self.p_updnb.data_ptr()
            pb.rmv_logits, pb.add_logits = self.p_rmv.data_ptr(), self.p_add.data_ptr()
            pb.slot_rows, pb.counters = self.p_slot_rows.data_ptr(), self.p_counters.data_ptr()
            pb.workspace, pb.workspace_bytes = self.p_ws.data_ptr(), self.p_ws.numel()
            pb.stats = self.d_stats.data_ptr()
            pb.row_cap = cap_rows
            self.p_big = torch.zeros((S, 2), dtype=torch.int32, device=dev)
            pb.slot_big = self.p_big.data_ptr()
            pb.rooms_have_pvox = 1 if self.have_pvox else 0
            self.packed_buffers = pb
            # free-running launches (lrg_grow_async): greedy growing through the single-launch front, lite 0 / 2
            can_free = ((self.G == 1 or self.speculate) and self.params.restarts == 1 and self.have_pvox and max(Ni, Nn) <= 512 and
                        getattr(self.net, 'lite', 0) != 1)
            if self.want_free_run and not can_free:
                raise ValueError('free-running launches need greedy growing (restarts = group_size = 1), rooms with packed voxel words, '
                                 'at most 512 + 512 points per set and lite 0 or 2')
            if self.want_free_run is None:
                # (auto: where a step is a chain of latencies -- up to ~100 slots; with hundreds of slots in flight the lock-step
                #  launches, whose tiles pack the rows of all slots, get more out of the chip: 272 rooms 1.20 M against 0.80 M)
                self.free_run = can_free and S <= _lib.LRG_FREE_RUN_AUTO_SLOTS and max(ns) <= _lib.LRG_FREE_RUN_AUTO_POINTS and os.environ.get('LRG_FREE_RUN', '1') != '0'
            else:
                self.free_run = bool(self.want_free_run)
            if self.free_run:
                qbytes = self.lib.lrg_grow_async_queue_bytes(S)
                self.a_queue = torch.zeros(qbytes // 4, dtype=torch.int32, device=dev)
                self.a_sync = torch.zeros((S, 16), dtype=torch.int32, device=dev)
                ab = LrgAsyncBuffers()
                ab.queue, ab.queue_bytes, ab.sync = self.a_queue.data_ptr(), qbytes, self.a_sync.data_ptr()
                ab.front_workgroups = self.free_run_fronts or int(os.environ.get('LRG_FREE_RUN_FRONTS', '0'))
                ab.teams = self.free_run_teams or int(os.environ.get('LRG_FREE_RUN_TEAMS', '0'))
                ab.compute_units = int(os.environ.get('LRG_FREE_RUN_CUS', '0'))
                self.fill_cus = int(os.environ.get('LRG_FREE_RUN_FILL_CUS', '0' if self.free_run_fill_cus is None else str(int(self.free_run_fill_cus))))
                self.main_stream = None
                if self.fill_cus > 0 and ab.compute_units == 0:
                    ab.compute_units = max(64, torch.cuda.get_device_properties(dev).multi_processor_count - self.fill_cus)
                    self.main_stream, self.fill_stream = fill_streams(dev, self.fill_cus)
                ab.poll_sleep = int(os.environ.get('LRG_FREE_RUN_POLL', '0'))
                ab.branch_parts = int(os.environ.get('LRG_FREE_RUN_PARTS', '0'))
                ab.gemv_units = self.free_run_units or int(os.environ.get('LRG_FREE_RUN_UNITS', '0'))          # -1: the pooled product as tasks of the tile teams
                ab.branch_waves = self.free_run_waves or int(os.environ.get('LRG_FREE_RUN_WAVES', '0'))
                ab.rows16 = 1 if (9 <= F <= 16 and os.environ.get('LRG_FREE_RUN_ROWS16', '1') != '0') else 0
                # LRG_FREE_RUN_POOL_ROWS=1: a branch tile leaves its column maxima as ONE row of 16-byte stores and the pooled-product units take
                # the maximum over a slot's tiles while loading (no atomicMax per column, no zeroing by the front workgroup).  Same labels;
                # measured 832 k against 850 k instance-steps/s at 68 rooms in flight (profiles/r04_pool_rows_ab.txt): each of the sixteen
                # units then loads ~3.5 rows per side instead of one pooled row, on the step's critical path -- off by default.
                if os.environ.get('LRG_FREE_RUN_POOL_ROWS', '0') == '1':
                    nb = self.lib.lrg_grow_async_pool_rows_bytes(ctypes.byref(self.net._w), S)
                    self.a_pool_rows = torch.zeros(nb // 4, dtype=torch.float32, device=dev)
                    ab.pool_rows, ab.pool_rows_bytes = self.a_pool_rows.data_ptr(), nb
                # in-launch fill-in (lrg_async.inl): finished rooms are filled in (:308-316) by tile teams of the same launch; the host only
                # reads the statistics block between two launches.  LRG_FREE_RUN_FILL=0: lrg_nn1_fill_batch between the launches, as in round 3.
                self.fill_in_launch = F == 13 and os.environ.get('LRG_FREE_RUN_FILL', '1') != '0' and getattr(self, 'fill_cus', 0) == 0
                if self.fill_in_launch:
                    self.a_fill_list = torch.zeros(tot, dtype=torch.int32, device=dev)
                    self.a_fill_best = torch.zeros(tot, dtype=torch.int64, device=dev)
                    self.a_fill_sync = torch.zeros((len(rooms), 4), dtype=torch.int32, device=dev)
                    ab.fill_list, ab.fill_best, ab.fill_sync = self.a_fill_list.data_ptr(), self.a_fill_best.data_ptr(), self.a_fill_sync.data_ptr()
                    ab.fill_label_base, ab.fill_out_base = self.d_label.data_ptr(), self.d_filled.data_ptr()
                    ab.fill_rooms = len(rooms)
                    ab.fill_wgs = int(os.environ.get('LRG_FREE_RUN_FILL_WGS', '0'))
                if self.tail_rows and ab.rows16:
                    tb = self.lib.lrg_grow_async_tail_bytes(S, self.tail_rows)
                    self.a_tail = torch.zeros(tb // 4 + 16, dtype=torch.int32, device=dev)
                    ab.tail_ctl = (self.a_tail.data_ptr() + 63) // 64 * 64
                    ab.tail_rows = self.tail_rows
                    ab.tail_close_us = int(os.environ.get('LRG_FREE_RUN_TAIL_US', '0'))
                self.a_work = torch.zeros(8, dtype=torch.int64, device=dev)      # evaluations, inlier rows, neighbour rows, tiles; speculation: regions voided, their evaluations
                ab.speculate = self.speculate if self.speculate > 1 else 0
                ab.work = self.a_work.data_ptr()
                if os.environ.get('LRG_FREE_RUN_DEBUG') == '1':          # stage-by-stage tick accumulators (tools/free_run_perf.py)
                    self.a_dbg = torch.zeros(64, dtype=torch.int64, device=dev)
                    ab.debug_ticks = self.a_dbg.data_ptr()
                self.async_buffers = ab
        self.h_stats = [torch.zeros(LRG_STATS_WORDS, dtype=torch.int64).pin_memory() for _ in range(self.depth)]
        self.ev = [torch.cuda.Event() for _ in range(self.depth)]
        self.group_room = [-1] * self.n_groups
        self._reset_pending = set()
        self.iterations = 0
        self._seen_done = 0
        self._polls = 0
        self._polls_seen = -1
        self._rooms_loaded = True
        return self

    # ------------------------------------------------------------------------------------------
    def reset_room(self, r):
        """Return room r to its pristine state (visited / labels cleared, cursor at 0)."""
        if getattr(self, 'fill_stream', None) is not None:      # (a fill-in of the room may still be reading its labels)
            torch.cuda.current_stream(self.dev).wait_stream(self.fill_stream)
        if self.rng == 'counter' and self.device_bind:
            self._reset_pending.add(r)          # folded into the device-side bind that follows (one launch, no upload)
            return
        o, n = int(self.room_off[r]), self.room_n[r]
        self.d_visited[o:o + n].zero_()
        self.d_label[o:o + n].zero_()
        R = self.h_rooms[r]
        R.next_cluster_id, R.seed_cursor, R.n_regions, R.done = 1, 0, 0, 0
        sz = ctypes.sizeof(LrgRoom)
        buf = np.frombuffer(bytes(R), dtype=np.uint8).copy()
        self.d_rooms[r * sz:(r + 1) * sz].copy_(torch.from_numpy(buf))

    def bind(self, group, r):
        """Bind slot group `group` to room r: every slot waits with seed -1, so the next lrg_advance
        picks the room's first seed (:186-188)."""
        if self.rng == 'counter' and self.device_bind:
            # on the device (lrg_bind_group): a pageable host-to-device copy would block the host until this lane's stream has
            # drained, and the other lanes would starve meanwhile
            reset = 1 if r in self._reset_pending else 0
            self._reset_pending.discard(r)
            _lib.check(self.lib.lrg_bind_group(_ptr(self.d_slots), _ptr(self.d_rooms), group * self.G, self.G, int(r), reset,
                                               1 if self.packed else 0, _stream_ptr(self.dev)), 'lrg_bind_group')
            self.group_room[group] = r
            return
        sz = ctypes.sizeof(LrgSlot)
        for s in range(group * self.G, (group + 1) * self.G):
            sl = self.h_slots[s]
            sl.room = r
            sl.status = LRG_WAIT if r >= 0 else LRG_IDLE
            sl.seed = -1
            sl.restart = sl.step = sl.steps_total = sl.stuck = 0
            sl.updated = -1
            sl.acc_add = sl.acc_rmv = -1
            sl.ml_score = sl.ml_best = 0.0
            sl.count = -1
            sl.best_count = -1
            sl.pad = 0
            sl.scan_cnt = 0
            sl.query = 0
            sl.spec_pos, sl.spec_flags = 2147483647, 0
            for d in range(3):
                sl.scan_mn[d], sl.scan_mx[d] = 2147483647, -2147483648
        a, b = group * self.G, (group + 1) * self.G
        buf = np.frombuffer(bytes(self.h_slots), dtype=np.uint8)[a * sz:b * sz].copy()
        self.d_slots[a * sz:b * sz].copy_(torch.from_numpy(buf))
        if self.packed:
            self.d_cur[a:b].zero_()
        self.group_room[group] = r

    def fill(self, r):
        """1-NN fill-in of room r's unlabeled points (test_region_grow.py:308-316) into d_filled.  Free-running launches with CUs
        left out for it (free_run_fill_cus): on the fill stream, beside the next launch -- the room was reported finished by a launch
        that has completed (poll_done), its labels are final; wait_fills() before reading d_filled."""
        if getattr(self, 'free_run', False) and getattr(self, 'fill_cus', 0) > 0 and not getattr(self, '_in_fill_stream', False):
            self._in_fill_stream = True
            try:
                with torch.cuda.stream(self.fill_stream):
                    self.fill(r)
            finally:
                self._in_fill_stream = False
            return
        o, n = int(self.room_off[r]), self.room_n[r]
        F = self.net.feature_size
        if getattr(self, '_fill_ws', None) is None:
            self._fill_ws = torch.empty(self.lib.lrg_nn1_fill_workspace_bytes(max(self.room_n)), dtype=torch.uint8, device=self.dev)
        _lib.check(self.lib.lrg_nn1_fill_ws(ctypes.c_void_p(self.d_points.data_ptr() + o * F * 4), n, F,
                                            ctypes.c_void_p(self.d_label.data_ptr() + o * 4),
                                            ctypes.c_void_p(self.d_filled.data_ptr() + o * 4), _ptr(self._fill_ws),
                                            self._fill_ws.numel(), _stream_ptr(self.dev)), 'lrg_nn1_fill_ws')

    def fill_many(self, rs):
        """fill(r) for every r of rs, the rooms filled in together (lrg_nn1_fill_batch: three launches per 64 rooms instead of four per
        room; the rooms that finish during one free-running launch)."""
        rs = list(rs)
        if getattr(self, 'free_run', False) and getattr(self, 'fill_cus', 0) > 0 and not getattr(self, '_in_fill_stream', False):
            self._in_fill_stream = True          # (CUs left out for the fill-ins: on the fill stream, as fill() does)
            try:
                with torch.cuda.stream(self.fill_stream):
                    self.fill_many(rs)
            finally:
                self._in_fill_stream = False
            return
        if len(rs) <= 1:
            for r in rs:
                self.fill(r)
            return
        F = self.net.feature_size
        jobs = (_lib.LrgFillJob * len(rs))()
        for k, r in enumerate(rs):
            o, n = int(self.room_off[r]), self.room_n[r]
            jobs[k].points = self.d_points.data_ptr() + o * F * 4
            jobs[k].label_in = self.d_label.data_ptr() + o * 4
            jobs[k].label_out = self.d_filled.data_ptr() + o * 4
            jobs[k].n = n
        need = self.lib.lrg_nn1_fill_batch_workspace_bytes(jobs, len(rs))
        if getattr(self, '_fill_ws', None) is None or self._fill_ws.numel() < need:
            self._fill_ws = torch.empty(max(need, self.lib.lrg_nn1_fill_workspace_bytes(max(self.room_n))), dtype=torch.uint8, device=self.dev)
        _lib.check(self.lib.lrg_nn1_fill_batch(jobs, len(rs), F, _ptr(self._fill_ws), self._fill_ws.numel(), _stream_ptr(self.dev)), 'lrg_nn1_fill_batch')

    def wait_fills(self):
        """Fill-ins enqueued on the fill stream are complete on return."""
        if getattr(self, 'fill_stream', None) is not None:
            self.fill_stream.synchronize()

    # ------------------------------------------------------------------------------------------
    def _release_graph(self):
        if getattr(self, '_graph', None):
            self.lib.lrg_step_graph_destroy(self._graph)
        self._graph = None

    def __del__(self):
        try:
            self._release_graph()
        except Exception:
            pass

    def _record_poll(self):
        k = self._polls % self.depth
        self.h_stats[k].copy_(self.d_stats, non_blocking=True)
        self.ev[k].record()
        self._polls += 1

    def enqueue_iteration(self):
        """One lock-step iteration, device-side randomness (no host sync)."""
        if self.packed:
            rc = self.lib.lrg_grow_step_packed(_ptr(self.d_slots), _ptr(self.d_rooms), self.S, self.cap, ctypes.byref(self.params),
                                               ctypes.byref(self.net._w), ctypes.byref(self.packed_buffers), _stream_ptr(self.dev))
            _lib.check(rc, 'lrg_grow_step_packed')
        else:
            flags = self.net.forward_flags | (_lib.LRG_FWD_POOL_ZEROED if self.net.mode == 'fused' else 0)
            rc = self.lib.lrg_grow_step(_ptr(self.d_slots), _ptr(self.d_rooms), self.S, self.cap, ctypes.byref(self.params),
                                        ctypes.byref(self.net._w), ctypes.byref(self.step_buffers), self.advance_rounds,
                                        flags, _stream_ptr(self.dev))
            _lib.check(rc, 'lrg_grow_step')
        self.iterations += 1
        if self.iterations % self.poll_every == 0:
            self._record_poll()

    def enqueue_graph(self):
        """`graph_iterations` packed iterations with one host call (HIP graph replay); the graph is captured on first use,
        on the stream that is current then, and replayed on the current stream."""
        if not self.packed or self.graph_iterations <= 0:
            raise _lib.LrgHipError('enqueue_graph needs packed iterations and graph_iterations > 0')
        st = torch.cuda.current_stream(self.dev)
        if self._graph is None:
            if st.cuda_stream == 0:
                raise _lib.LrgHipError('a HIP graph cannot be captured on the null stream: enter torch.cuda.stream(...) first')
            g = ctypes.c_void_p()
            rc = self.lib.lrg_step_graph_create(_ptr(self.d_slots), _ptr(self.d_rooms), self.S, self.cap, ctypes.byref(self.params),
                                                ctypes.byref(self.net._w), ctypes.byref(self.packed_buffers), self.graph_iterations,
                                                ctypes.c_void_p(st.cuda_stream), ctypes.byref(g))
            _lib.check(rc, 'lrg_step_graph_create')
            self._graph = g
        _lib.check(self.lib.lrg_step_graph_launch(self._graph, ctypes.c_void_p(st.cuda_stream)), 'lrg_step_graph_launch')
        self.iterations += self.graph_iterations
        self._record_poll()

    def free_run_breakdown(self, since=None):
        """Stage-by-stage microseconds of the free-running launches so far (LRG_FREE_RUN_DEBUG=1 and a -DLRG_ASYNC_DEBUG=1 build; layout:
        csrc/lrg_async.inl, LrgAsyncArgs.dbg), or None.  `since`: an earlier return value of free_run_ticks() to take the difference to."""
        d = self.free_run_ticks()
        if d is None:
            return None
        if since is not None:
            d = d - since
        ev = max(d[6], 1.0)
        row = {'us': dict(front_busy_per_step=d[0] / max(d[1], 1) / 100, last_branch_tile_in=d[2] / ev / 100, last_pooled_block_in=d[3] / ev / 100,
                          last_head_tile_in=d[4] / ev / 100, seen_by_front=d[5] / ev / 100,
                          branch_tile=d[10] / max(d[11], 1) / 100, pooled_block=d[12] / max(d[13], 1) / 100, head_tile=d[14] / max(d[15], 1) / 100,
                          team_wait_per_task=d[16] / max(d[17], 1) / 100),
               'front_phase_us': dict(zip(['update', 'commit_seed', 'query', 'sampling', 'gather(+small medians)', 'big medians', 'gather alone', 'one median alone'],
                                          [float(x) / max(d[1], 1) / 100 for x in list(d[21:28]) + [d[20]]])),
               'tasks_per_evaluation': dict(branch=d[11] / ev, pooled=d[13] / ev, head=d[15] / ev), 'evaluations': float(d[6]), 'front_steps': float(d[1]),
               'front_looks_in_vain_per_evaluation': float(d[28]) / ev,
               'seen_after_last_head_tile': dict(mean_us=d[62] / ev / 100, histogram_us=dict(zip(['<0.5', '<1', '<2', '<4', '<8', '<16', '<32', '>=32'], [float(x) / ev for x in d[54:62]])))}
        if d[32] > 0:      # LRG_TRACE build: mean cycles since the tile began at each stamp
            names = ['staged'] + [x for l in range(5) for x in ('L%d start' % l, 'L%d end' % l)] + \
                    [x for c in range(4) for x in ('pass%d mfma' % c, 'pass%d epilogue' % c)] + ['end']
            row['tile_cycles'] = {n: int(d[33 + i] / d[32]) for i, n in enumerate(names) if d[33 + i] > 0}
        return row

    def free_run_ticks(self):
        import numpy as _np
        if getattr(self, 'a_dbg', None) is None:
            return None
        return self.a_dbg.cpu().numpy().astype(_np.float64)

    def enqueue_free_run(self, steps=None, budget_us=None):
        """One free-running launch: every slot up to `steps` grow steps at its own pace (lrg_grow_async)."""
        steps = self.free_run_steps if steps is None else int(steps)
        budget = self.free_run_budget_us if budget_us is None else int(budget_us)
        rc = self.lib.lrg_grow_async(_ptr(self.d_slots), _ptr(self.d_rooms), self.S, self.cap, ctypes.byref(self.params),
                                     ctypes.byref(self.net._w), ctypes.byref(self.packed_buffers), ctypes.byref(self.async_buffers),
                                     steps, budget, _stream_ptr(self.dev))
        _lib.check(rc, 'lrg_grow_async')
        self.launches = getattr(self, 'launches', 0) + 1      # (`iterations` counts lock-step iterations only; the steps taken: instance_steps)
        self._record_poll()

    def enqueue(self):
        """The next batch of iterations by the cheapest route: a free-running launch, a graph replay when one is configured, else
        one iteration."""
        if self.free_run:
            self.enqueue_free_run()
        elif self.packed and self.graph_iterations > 0 and torch.cuda.current_stream(self.dev).cuda_stream != 0:
            self.enqueue_graph()
        else:
            self.enqueue_iteration()

    def poll_done(self, wait=False):
        """Groups whose room finished, as seen `depth-1` read-backs ago (or at the latest one, if wait)."""
        self.done_rooms = []          # room of each finished group as the device saw it (greedy front kernels: slot | room << 32)
        self.done_filled = []         # ... and whether the launch that finished it also filled it in
        if self._polls == 0:
            return []
        if wait:
            k = (self._polls - 1) % self.depth
        else:
            if self._polls < self.depth or self._polls == self._polls_seen:
                return []
            k = self._polls % self.depth
        self._polls_seen = self._polls
        self.ev[k].synchronize()
        st = self.h_stats[k]
        done_total = int(st[1])
        out = []
        if done_total - self._seen_done > LRG_DONE_RING:
            raise _lib.LrgHipError('done ring overflow')
        for j in range(self._seen_done, done_total):
            e = int(st[4 + (j % LRG_DONE_RING)])
            out.append((e & 0x7FFFFFFF) // self.G)
            self.done_rooms.append(e >> 32)
            self.done_filled.append(bool(e & 0x80000000))      # (a free-running launch filled the room in itself: LrgAsyncBuffers.fill_list)
        self._seen_done = done_total
        self.last_stats = (int(st[0]), int(st[1]), int(st[2]))
        if int(st[3]):
            raise _lib.LrgHipError('lrg_grow_async gave up on a hand-over between workgroups (%d front workgroups; sum of their reasons %d -- 1 launch '
                                   'past its time limit, 2 a team waited too long for a task, 3 a team lost a wavefront at a barrier, 4 / 5 a pooled-product '
                                   'unit / a head tile waited too long, 6 not all workgroups of the launch were resident within its budget + 20 ms: something else holds '
                                   'compute units of this device, 7 the fill-in ring stayed full): results are invalid'
                                   % (int(st[3]) & 0xFFFFFFFF, int(st[3]) >> 32))
        return out

    # ------------------------------------------------------------------------------------------
    def reset_state(self):
        """Forget the read-back history (a fresh pass over loaded rooms): the done ring restarts from the device's current count."""
        torch.cuda.current_stream(self.dev).synchronize()
        st = self.d_stats.cpu()
        self._seen_done = int(st[1])
        self._polls = 0
        self._polls_seen = -1

    def set_room_queue(self, rooms, reset=False):
        """Free-running launches: the rooms (indices) that wait for a slot.  A slot whose room is finished takes the next one inside
        the launch (LrgAsyncBuffers.room_queue); the host only learns of finished rooms (poll_done: done_rooms) and fills them in."""
        q = np.zeros(2 + max(1, len(rooms)), dtype=np.int32)
        q[1] = len(rooms)
        q[2:2 + len(rooms)] = np.asarray(rooms, dtype=np.int64) | ((1 << 30) if reset else 0)
        self.a_roomq = torch.from_numpy(q).to(self.dev)
        self.async_buffers.room_queue = self.a_roomq.data_ptr()

    def room_order(self):
        """The order in which the loaded rooms take slots: as loaded while they are few; with more than two rounds of slots, long rooms early and sizes mixed
        (dist.queue_order -- rooms are independent and the random stream is keyed by the room, so the labels do not depend on it)."""
        if self.n_rooms <= 2 * self.n_groups or self.room_order_mode == 'loaded' or os.environ.get('LRG_ROOM_ORDER', '') == 'loaded':
            return list(range(self.n_rooms))
        from .dist import queue_order
        return queue_order(list(range(self.n_rooms)), [int(n) for n in self.room_n[:self.n_rooms]], self.n_groups)

    def free_run_begin(self):
        """Free-running launches over ALL loaded rooms: the first S rooms bound by the host, the rest handed out on the device."""
        self.reset_state()
        order = self.room_order()
        first = min(self.n_groups, self.n_rooms)
        for g in range(self.n_groups):
            self.bind(g, order[g] if g < first else -1)
        self.set_room_queue(order[first:])
        self.rooms_finished = 0

    def verify_fills_in_launch(self):
        """A room the done ring reported as filled in by its launch (bit 31) must have its 'filled' word set by the last fill task
        (LrgAsyncBuffers.fill_sync[room][3]); a room whose tasks did not complete is filled in here, by the host-launched kernels --
        never left with stale labels.  Returns the rooms that needed it (normally none)."""
        rooms = sorted(set(getattr(self, '_filled_in_launch', [])))
        self._filled_in_launch = []
        if not rooms or not getattr(self, 'fill_in_launch', False):
            return []
        torch.cuda.current_stream(self.dev).synchronize()
        flags = self.a_fill_sync[:, 3].cpu().numpy()
        missing = [r for r in rooms if flags[r] != 1]
        if missing:
            self.fill_many(missing)
            torch.cuda.current_stream(self.dev).synchronize()
        self.fills_redone = getattr(self, 'fills_redone', 0) + len(missing)
        return missing

    def free_run_step(self, fill=True, steps=None, budget_us=None, wait=False):
        """One launch, then the fill-in (test_region_grow.py:308-316) of the rooms reported finished since the last call; returns
        how many those were."""
        self.enqueue_free_run(steps, budget_us)
        self.poll_done(wait=wait)
        n = len(self.done_rooms)
        # (the last rooms of a pass: no launch follows that their fill-ins could run beside -- on the launches' stream, with the whole chip)
        last = getattr(self, 'fill_cus', 0) > 0 and self.rooms_finished + n >= self.n_rooms
        if last:
            self.wait_fills()
            self._in_fill_stream = True
        if fill:
            self.fill_many([r for r, f in zip(self.done_rooms, self.done_filled) if not f])
            if not hasattr(self, '_filled_in_launch'):
                self._filled_in_launch = []
            self._filled_in_launch.extend(r for r, f in zip(self.done_rooms, self.done_filled) if f)
        if last:
            self._in_fill_stream = False
        self.rooms_finished += n
        if getattr(self, '_trace_done', None) is not None:
            self._trace_done.extend(int(r) for r in self.done_rooms)
        self.done_rooms = []
        return n

    def _grow_loaded_free_run(self, fill=True):
        main = getattr(self, 'main_stream', None)
        if main is not None and torch.cuda.current_stream(self.dev).cuda_stream != main.cuda_stream:
            # CUs left out for the fill-ins: the launches go to the stream confined to the rest (fill_streams)
            main.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(main):
                return self._grow_loaded_free_run(fill)
        self.free_run_begin()
        trace = [] if os.environ.get('LRG_FREE_RUN_TRACE') == '1' else None      # (per launch: seconds, rooms reported finished, steps the device had counted)
        self._trace_done = [] if trace is not None else None
        t0 = time.perf_counter()
        while self.rooms_finished < self.n_rooms:
            self.free_run_step(fill)
            if trace is not None:
                trace.append((round(time.perf_counter() - t0, 4), self.rooms_finished, getattr(self, 'last_stats', (0, 0, 0))[2]))
        torch.cuda.current_stream(self.dev).synchronize()
        if trace is not None:
            import sys
            sys.stderr.write('LRG_FREE_RUN_TRACE %d slots %d rooms: %s\n' % (self.S, self.n_rooms, trace))
            sys.stderr.write('LRG_FREE_RUN_LAST %d slots: %s\n' % (self.S, [(r, int(self.room_n[r])) for r in self._trace_done[-60:]]))      # (queue position, points)
        self.wait_fills()
        if fill:
            self.verify_fills_in_launch()
        self.async_buffers.room_queue = None
        return self.n_rooms

    def grow_loaded(self, fill=True):
        """Counter stream: grow (and fill in) every loaded room from its current state; labels final on the device on return."""
        assert self.rng == 'counter'
        if self.free_run:
            return self._grow_loaded_free_run(fill)
        self.reset_state()
        queue = self.room_order()
        for g in range(self.n_groups):
            self.bind(g, queue.pop(0) if queue else -1)
        finished = 0
        while finished < self.n_rooms:
            self.enqueue()
            gs = self.poll_done()
            if fill:
                self.fill_many([self.group_room[g] for g in gs])
            for g in gs:
                finished += 1
                self.bind(g, queue.pop(0) if queue else -1)
        torch.cuda.current_stream(self.dev).synchronize()
        return self.n_rooms

    def run_timed(self, rooms, fill=True):
        """One room at a time (the reference's own schedule) with HIP events round the two halves of every iteration, for the
        reference's timing buckets (test_region_grow.py:40-51): 'net' = the LrgNet launches; the front kernel's time is split into
        'inlier' (mask update, bounding box, stop decision, commit -- :260-306) and 'neighbor' (box query, median, sampling,
        stacking -- :219-254) in proportion to the wall-clock ticks its workgroup spent in either part.
        Returns (results, buckets): buckets[r] = dict(neighbor, net, inlier in seconds, iter_* = per-iteration lists)."""
        assert self.rng == 'counter' and self.n_groups == 1 and self.G == 1
        results, buckets = [], []
        with torch.cuda.device(self.dev):
            for room in rooms:
                self.load_rooms([room])
                if not (self.packed and self.have_pvox):
                    raise _lib.LrgHipError('run_timed needs the packed greedy iteration')
                ticks = torch.zeros((self.S, 2), dtype=torch.int64, device=self.dev)
                self.packed_buffers.phase_ticks = ticks.data_ptr()
                self.reset_state()
                self.bind(0, 0)
                st = _stream_ptr(self.dev)
                pb, evs = self.packed_buffers, []
                while True:
                    for _ in range(32):
                        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                        e[0].record()
                        _lib.check(self.lib.lrg_front_step(_ptr(self.d_slots), _ptr(self.d_rooms), self.S, self.cap, ctypes.byref(self.params),
                                                           ctypes.byref(self.net._w), ctypes.byref(pb), st), 'lrg_front_step')
                        e[1].record()
                        _lib.check(self.lib.lrg_forward_packed(ctypes.byref(self.net._w), pb.x_in, pb.x_nb,
                                                               self.lib.lrg_packed_rows_center(ctypes.byref(self.params), ctypes.byref(pb)),
                                                               pb.row_slot_in, pb.row_slot_nb,
                                                               pb.counters, ctypes.c_void_p(pb.counters + 8), self.S, pb.row_cap, pb.add_logits,
                                                               pb.rmv_logits, pb.workspace, pb.workspace_bytes, _lib.LRG_FWD_POOL_ZEROED, st),
                                   'lrg_forward_packed')
                        e[2].record()
                        evs.append(e)
                        self.iterations += 1
                    torch.cuda.current_stream(self.dev).synchronize()
                    if int(self.d_stats[1].item()) > self._seen_done:
                        break
                if fill:
                    self.fill(0)
                torch.cuda.current_stream(self.dev).synchronize()
                front = np.array([e[0].elapsed_time(e[1]) for e in evs]) * 1e-3
                netw = np.array([e[1].elapsed_time(e[2]) for e in evs]) * 1e-3
                tk = ticks.cpu().numpy()[0].astype(np.float64)
                share = tk[1] / max(tk.sum(), 1.0)
                buckets.append(dict(neighbor=float(front.sum() * share), net=float(netw.sum()), inlier=float(front.sum() * (1.0 - share)),
                                    iter_neighbor=(front * share).tolist(), iter_net=netw.tolist(), iter_inlier=(front * (1.0 - share)).tolist()))
                self.packed_buffers.phase_ticks = None
                results.append(self.collect(fill)[0])
        return results, buckets

    def _legacy_iteration(self, streams):
        lib, st, P = self.lib, _stream_ptr(self.dev), ctypes.byref(self.params)
        S, Ni, Nn = self.S, self.net.num_inlier_points, self.net.num_neighbor_points
        _lib.check(lib.lrg_bbox_stop(_ptr(self.d_slots), _ptr(self.d_rooms), S, self.cap, P, st), 'lrg_bbox_stop')
        for _ in range(self.advance_rounds):
            _lib.check(lib.lrg_advance(_ptr(self.d_slots), _ptr(self.d_rooms), S, P, _ptr(self.d_stats), st), 'lrg_advance')
            _lib.check(lib.lrg_box_query(_ptr(self.d_slots), _ptr(self.d_rooms), S, self.cap, P, st), 'lrg_box_query')
        slots = self._read_slots()
        active = [s for s in range(S) if slots[s].status == LRG_ACTIVE]
        if active:
            sin = np.zeros((S, Ni), dtype=np.int32)
            snb = np.zeros((S, Nn), dtype=np.int32)
            for s in active:
                rs = streams[self.group_room[s // self.G]]
                nc, ne = slots[s].nc, slots[s].ne
                # test_region_grow.py:237-240 / :249-252, same calls in the same order
                sin[s] = rs.choice(nc, Ni, replace=False) if nc >= Ni else \
                    list(range(nc)) + list(rs.choice(nc, Ni - nc, replace=True))
                snb[s] = rs.choice(ne, Nn, replace=False) if ne >= Nn else \
                    list(range(ne)) + list(rs.choice(ne, Nn - ne, replace=True))
            self.b_sin.copy_(torch.from_numpy(sin))
            self.b_snb.copy_(torch.from_numpy(snb))
            _lib.check(lib.lrg_median(_ptr(self.d_slots), _ptr(self.d_rooms), S, P, _ptr(self.b_center), st), 'lrg_median')
            _lib.check(lib.lrg_gather_center(_ptr(self.d_slots), _ptr(self.d_rooms), S, P, _ptr(self.b_sin),
                                             _ptr(self.b_snb), _ptr(self.b_center), _ptr(self.b_inl), _ptr(self.b_nbr),
                                             _ptr(self.b_gtr), _ptr(self.b_gta), None, None, st), 'lrg_gather_center')
            self.net.forward(self.b_inl, self.b_nbr, self.b_add, self.b_rmv)
            amask = rmask = None
            need_logits = self.policy != 'gt'
            if need_logits:
                add = self.b_add.cpu().numpy()
                rmv = self.b_rmv.cpu().numpy()
            am = np.zeros((S, Nn), dtype=np.uint8)
            rm = np.zeros((S, Ni), dtype=np.uint8)
            for s in active:
                rs = streams[self.group_room[s // self.G]]
                u_add = rs.random_sample(Nn)                                                # :266
                u_rmv = rs.random_sample(Ni)                                                # :267
                if need_logits:
                    add_conf, rmv_conf = _softmax_conf(add[s]), _softmax_conf(rmv[s])       # :262-263
                    if self.policy == 'net':
                        am[s], rm[s] = u_add < add_conf, u_rmv < rmv_conf
                    else:
                        am[s], rm[s] = add_conf > 0.5, rmv_conf > 0.5
            if need_logits:
                self.b_amask.copy_(torch.from_numpy(am))
                self.b_rmask.copy_(torch.from_numpy(rm))
                amask, rmask = self.b_amask, self.b_rmask
            if self.debug_hook is not None:
                center = self.b_center.cpu().numpy()
                inl, nbr = self.b_inl.cpu().numpy(), self.b_nbr.cpu().numpy()
                cur = self.d_cur.cpu().numpy()
                for s in active:
                    r = self.group_room[s // self.G]
                    self.debug_hook(dict(slot=s, room=r, seed=slots[s].seed, restart=slots[s].restart, step=slots[s].step,
                                         nc=slots[s].nc, ne=slots[s].ne, center=center[s].copy(), inlier=inl[s].copy(),
                                         neighbor=nbr[s].copy(), subset_in=sin[s].copy(), subset_nb=snb[s].copy(),
                                         add=add[s].copy() if need_logits else None,
                                         rmv=rmv[s].copy() if need_logits else None,
                                         add_mask=am[s].astype(bool), rmv_mask=rm[s].astype(bool),
                                         mask_before=cur[s, :self.room_n[r]].astype(bool),
                                         min_dims=np.array(slots[s].mn[:]), max_dims=np.array(slots[s].mx[:])))
            prm = LrgGrowParams.from_buffer_copy(self.params)
            if self.policy == 'threshold':
                prm.policy = 0      # the masks were thresholded on the host
            _lib.check(lib.lrg_mask_update(_ptr(self.d_slots), _ptr(self.d_rooms), S, ctypes.byref(prm), _ptr(self.b_inl),
                                           _ptr(self.b_nbr), _ptr(self.b_center), _ptr(self.b_add), _ptr(self.b_rmv),
                                           _ptr(self.b_gtr), _ptr(self.b_gta), _ptr(amask), _ptr(rmask), None, None,
                                           _ptr(self.d_stats), st), 'lrg_mask_update')
        self.iterations += 1
        return slots

    def _read_slots(self):
        raw = self.d_slots.cpu().numpy().tobytes()
        return (LrgSlot * self.S).from_buffer_copy(raw)

    def _read_rooms(self):
        raw = self.d_rooms.cpu().numpy().tobytes()
        return (LrgRoom * self.n_rooms).from_buffer_copy(raw)

    # ------------------------------------------------------------------------------------------
    def run(self, rooms, fill=True, max_iterations=None, legacy_seeds=None, legacy_shared_seed=None):
        """Grow every room once; returns a RoomResult per room (in input order).
        rng='legacy': one numpy.random.RandomState per room, seeded with legacy_seeds[r] (default: the room id).  The reference
        script seeds ONCE (numpy.random.seed(0), test_region_grow.py:21) and draws all rooms of a file from that one stream in
        file order: legacy_shared_seed=0 with rooms_in_flight=1 reproduces exactly that."""
        with torch.cuda.device(self.dev):
            return self._run(rooms, fill, max_iterations, legacy_seeds, legacy_shared_seed)

    def _run(self, rooms, fill, max_iterations, legacy_seeds, legacy_shared_seed=None):
        self.load_rooms(rooms)
        if self.rng != 'legacy' and self.free_run and not max_iterations:
            self._grow_loaded_free_run(fill)          # (binds the first rooms itself, the rest wait in the device-side queue)
            torch.cuda.synchronize()
            return self.collect(fill)
        queue = list(range(self.n_rooms)) if self.rng == 'legacy' else self.room_order()      # (legacy streams are consumed in the loaded order: the reference's)
        for g in range(self.n_groups):
            self.bind(g, queue.pop(0) if queue else -1)
        finished = 0
        if self.rng == 'legacy':
            if legacy_shared_seed is not None:
                if self.n_groups != 1:
                    raise ValueError('one shared stream is consumed room after room: rooms_in_flight must be 1')
                streams = [np.random.RandomState(legacy_shared_seed)] * self.n_rooms
            else:
                seeds = legacy_seeds if legacy_seeds is not None else [self.room_ids[r] for r in range(self.n_rooms)]
                streams = [np.random.RandomState(sd) for sd in seeds]
            while finished < self.n_rooms:
                slots = self._legacy_iteration(streams)
                for g in range(self.n_groups):
                    r = self.group_room[g]
                    if r >= 0 and slots[g * self.G].status == LRG_DONE:
                        if fill:
                            self.fill(r)
                        finished += 1
                        self.bind(g, queue.pop(0) if queue else -1)
                if max_iterations and self.iterations >= max_iterations:
                    break
        else:
            # (max_iterations counts lock-step iterations: a free-running grower takes them one by one too -- the two formulations
            #  share their buffers and give the same results -- instead of one launch of up to free_run_steps steps per slot)
            step = self.enqueue_iteration if (self.free_run and max_iterations) else self.enqueue
            while finished < self.n_rooms:
                step()
                gs = self.poll_done()
                if fill:
                    self.fill_many([self.group_room[g] for g in gs])
                for g in gs:
                    finished += 1
                    self.bind(g, queue.pop(0) if queue else -1)
                if max_iterations and self.iterations >= max_iterations:
                    break
        torch.cuda.synchronize()
        return self.collect(fill)

    def collect(self, fill=True):
        self.wait_fills()
        rooms = self._read_rooms()
        label = self.d_label.cpu().numpy()
        filled = self.d_filled.cpu().numpy() if fill else None
        rlog = self.d_rlog.cpu().numpy()
        out = []
        for r in range(self.n_rooms):
            o, n = int(self.room_off[r]), self.room_n[r]
            regs = []
            for k in range(rooms[r].n_regions):
                row = rlog[o + k]
                regs.append(dict(seed=int(row[0]), steps=int(row[1]), points=int(row[2]),
                                 reason=REASON_NAMES.get(int(row[3]), str(int(row[3]))), labeled=bool(row[4]),
                                 restart=int(row[5]),
                                 # add_acc / remove_acc of the region's last evaluated step (learn_region_grow_util.py:175,180)
                                 add_acc=(int(row[6]) / float(self.net.num_neighbor_points)) if row[6] >= 0 else float('nan'),
                                 rmv_acc=(int(row[7]) / float(self.net.num_inlier_points)) if row[7] >= 0 else float('nan')))
            out.append(RoomResult(self.room_ids[r], label[o:o + n].astype(np.int64),
                                  filled[o:o + n].astype(np.int64) if fill else None, regs))
        return out

    @property
    def instance_steps(self):
        return int(self.d_stats[2].item())


def auto_lanes(slots_in_flight):
    """Lanes by the number of slots in flight: 1 below 24, 2 below 64, 3 from there.  At a fixed number of rooms in flight lanes buy
    little: the loop is a chain of latency-bound launches whose durations hardly depend on the slot count, so a lane with a third of
    the slots has a third of the throughput; what more lanes add is one lane's launches in the shadow of another's.  One MI355X,
    since a front workgroup has its CU to itself and the loop's tiles share a CU two at most (csrc/lrg_front.inl, lrg_fused.hip) --
    68 rooms: 1 / 2 / 3 / 4 lanes 560.6 / 597.2 / 607.9 / 348 k instance-steps/s (before: 531 / 557 / ~500 k); 39 ScanNet-shaped rooms:
    385 / 397 / 396 k; 32 rooms: 322 / 335 k; 272 rooms: 2 / 3 lanes 1.147 / 1.166 M; 16 restarts per seed (1088 slots): 1.132 / 1.192 M
    (tools/r02_lanes3.sh, r02_lanes4.sh)."""
    if slots_in_flight < 24:
        return 1
    return 2 if slots_in_flight < 64 else 3


_LANE_STREAMS = {}       # (device index, CU-masked lane count or 0) -> ([torch streams], [raw handles]), one set per process


_FILL_STREAMS = {}


def fill_streams(device, fill_cus):
    """-> (launch stream, fill stream) of `device`, process-wide: the first confined to all CUs but `fill_cus` of them, the second to those
    (lrg_stream_create_cu_mask).  A free-running launch holds every CU it is given for its whole duration; on plain streams a kernel of
    another stream is only placed beside it when EVERY shader engine has a CU to spare (32 CUs of an MI355X, tools/r03_side_stream.py);
    with the two streams masked to disjoint sets the fill-ins run on their few CUs while the launch runs on the rest
    (tools/r03_masked_streams.py: CUs 0 .. 7 of the mask's numbering work, one CU in eight does not)."""
    device = torch.device(device)
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev_index, int(fill_cus))
    if key not in _FILL_STREAMS:
        lib = _lib.load()
        ncu = torch.cuda.get_device_properties(device).multi_processor_count
        words = (ncu + 31) // 32
        out = []
        with torch.cuda.device(device):
            for cus in (range(int(fill_cus), ncu), range(int(fill_cus))):
                mask = (ctypes.c_uint32 * words)()
                for b in cus:
                    mask[b // 32] |= 1 << (b % 32)
                h = ctypes.c_void_p()
                _lib.check(lib.lrg_stream_create_cu_mask(mask, words, ctypes.byref(h)), 'lrg_stream_create_cu_mask')
                out.append((h, torch.cuda.ExternalStream(h.value, device=device)))
        _FILL_STREAMS[key] = out
    return _FILL_STREAMS[key][0][1], _FILL_STREAMS[key][1][1]


def lane_streams(device, lanes, cu_partition=False):
    """-> the process-wide streams of lanes 0 .. lanes-1 on `device`.

    One set per process, created once: HIP deals new streams round its hardware queues, and not every pair of queues runs
    side by side (tools/stream_overlap.hip: two kernels at most are in flight; a second pair of streams created later for the
    fixed-work leg of bench.py took turns instead: 207 rooms/s against 357 with the first pair reused).
    cu_partition: every lane's stream confined to its own 1/lanes of the compute units (lrg_stream_create_cu_mask; measured:
    no gain -- lanes do not compete for CUs -- so off by default)."""
    device = torch.device(device)
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev_index, lanes if cu_partition and lanes > 1 else 0)
    streams, raw = _LANE_STREAMS.setdefault(key, ([], []))
    if key[1] == 0:
        while len(streams) < lanes:
            streams.append(torch.cuda.Stream(device=device))
        return streams[:lanes]
    if not streams:
        lib = _lib.load()
        ncu = torch.cuda.get_device_properties(device).multi_processor_count
        words = (ncu + 31) // 32
        with torch.cuda.device(device):
            for k in range(lanes):
                lo, hi = k * ncu // lanes, (k + 1) * ncu // lanes
                mask = (ctypes.c_uint32 * words)()
                for b in range(lo, hi):
                    mask[b // 32] |= 1 << (b % 32)
                h = ctypes.c_void_p()
                _lib.check(lib.lrg_stream_create_cu_mask(mask, words, ctypes.byref(h)), 'lrg_stream_create_cu_mask')
                raw.append(h)
                streams.append(torch.cuda.ExternalStream(h.value, device=device))
    return streams[:lanes]


def auto_speculate(rooms_in_flight):
    """Regions of one room in flight (RegionGrower(speculate=K)) by the number of rooms this GPU holds at a time.  A room is a chain of dependent
    steps (test_region_grow.py:186-188); with few rooms the chip idles through it, and a front workgroup (one per room) serves ~3 slots before its own
    ~25 us per step become the bound.  Measured (profiles/r05_speculation.txt): see DESIGN.md section 3.0."""
    n = int(rooms_in_flight)
    # (profiles/r05_speculation.txt: one 100 k-point scene 1.60 x at K = 3 (2 / 4 / 6: 1.51 / 1.56 / 1.42), eight scenes 127 k -> 181 / 193 / 189 k kept steps/s at
    #  K = 2 / 3 / 4; 16 Area-5-shaped rooms 285 -> 340 k at K = 3; 68 rooms: K = 2 loses 7 % -- the chip is busy there without it)
    return 3 if n <= 16 else 2 if n <= 32 else 0


class LanedRegionGrower:
    """The rooms in flight dealt over `lanes` RegionGrower instances, each on its own HIP stream.

    Within one lane an iteration is a strict chain: a dozen small, latency-bound loop kernels, then the LrgNet evaluation that
    fills the chip, then the mask update.  Two lanes half a batch each let one lane's loop kernels run in the shadow of the
    other's network evaluation (+8 % instance-steps/s at 68 rooms on one MI355X).  Rooms are independent and the counter
    random stream is keyed by room id, so results do not depend on the lane count (tests/test_gpu_grow.py)."""

    def __init__(self, net, rooms_in_flight=64, lanes=None, cu_partition=False, **kw):
        if kw.get('rng', 'counter') != 'counter':
            raise ValueError("lanes need rng='counter' (the legacy stream is replayed on the host, one iteration at a time)")
        self.net = net
        self._build(rooms_in_flight, lanes, cu_partition, kw)

    def _build(self, rooms_in_flight, lanes, cu_partition, kw):
        """Lane count: given, or automatic (decided per room set in load_rooms; until rooms are seen: a single free-running lane where the
        arguments allow one, else auto_lanes)."""
        self._auto = None
        if lanes is None or int(lanes) <= 0:
            lockstep_lanes = auto_lanes(int(rooms_in_flight) * int(kw.get('restarts', 1)))
            lanes = lockstep_lanes
            # free-running launches (RegionGrower's choice up to LRG_FREE_RUN_AUTO_SLOTS greedy slots) fill the chip by themselves: one lane
            if (kw.get('free_run', None) is not False and int(kw.get('restarts', 1)) == 1 and int(rooms_in_flight) <= _lib.LRG_FREE_RUN_AUTO_SLOTS and
                    kw.get('packed', None) is not False and os.environ.get('LRG_FREE_RUN', '1') != '0'):
                lanes = 1
                if lockstep_lanes > 1:
                    self._auto = dict(rooms_in_flight=rooms_in_flight, lockstep_lanes=lockstep_lanes, cu_partition=cu_partition, kw=dict(kw))
        self._build_lanes(rooms_in_flight, lanes, cu_partition, kw)

    def _build_lanes(self, rooms_in_flight, lanes, cu_partition, kw):
        net = self.net
        lanes = max(1, min(int(lanes), int(rooms_in_flight)))
        if lanes > 1 and kw.get('free_run', None):
            # a free-running launch needs ALL its workgroups resident at once (one per CU: they wait for each other); a second one on another
            # stream finds the CUs taken and gives up at its start rendezvous (reason 6) -- refused here instead
            raise ValueError('free_run=True needs lanes=1: a free-running launch occupies every compute unit of the device (DESIGN.md section 4)')
        if lanes > 1:
            kw = dict(kw, free_run=False)      # lanes are lock-step growers side by side
        share = [rooms_in_flight // lanes + (1 if k < rooms_in_flight % lanes else 0) for k in range(lanes)]
        self.streams = lane_streams(net.device, lanes, cu_partition)
        self.growers = []
        for k in range(lanes):
            with torch.cuda.stream(self.streams[k]):
                self.growers.append(RegionGrower(net, rooms_in_flight=share[k], **kw))
        self.where = []          # original room index -> (lane, index within the lane)

    def load_rooms(self, rooms):
        if self._auto is not None and rooms:
            # lanes chosen automatically: ONE free-running lane where free-running launches apply to these rooms, else lock-step lanes as
            # auto_lanes has them -- decided from the room list (RegionGrower.free_run_applies) BEFORE any grower is built or loaded, and
            # decided again for every room set
            a = self._auto
            free = RegionGrower.free_run_applies(self.net, rooms, a['rooms_in_flight'], **a['kw'])
            want = 1 if free else a['lockstep_lanes']
            if want != len(self.growers) or (not free and any(g.want_free_run is not False for g in self.growers)):
                for gr in self.growers:
                    gr._release_graph()
                self.growers = []
                torch.cuda.empty_cache()
                self._build_lanes(a['rooms_in_flight'], want, a['cu_partition'], a['kw'] if free else dict(a['kw'], free_run=False))
        L = len(self.growers)
        order = sorted(range(len(rooms)), key=lambda i: -len(rooms[i]['points']))     # largest first, dealt round the lanes
        parts = [[] for _ in range(L)]
        self.where = [None] * len(rooms)
        for j, i in enumerate(order):
            k = j % L
            self.where[i] = (k, len(parts[k]))
            room = dict(rooms[i])
            room.setdefault('room_id', i)
            parts[k].append(room)
        if L == 1 and rooms:
            gr = self.growers[0]
            with torch.cuda.stream(self.streams[0]):
                gr.load_rooms(parts[0])
            gr.room_index = list(order)
            torch.cuda.synchronize()
            return
        for k, gr in enumerate(self.growers):
            gr.n_rooms = 0
            gr.room_index = [i for i in range(len(rooms)) if self.where[i][0] == k]      # input index of the lane's rooms,
            gr.room_index.sort(key=lambda i: self.where[i][1])                            #   in the lane's order
            if parts[k]:
                with torch.cuda.stream(self.streams[k]):
                    gr.load_rooms(parts[k])
        torch.cuda.synchronize()

    def run(self, rooms, fill=True):
        """Grow every room once; RoomResults in input order."""
        self.load_rooms(rooms)
        self.grow_loaded(fill)
        return self.collect(fill)

    def grow_loaded(self, fill=True):
        """Grow (and fill in) every loaded room, device-resident from start to end: on return the labels are final in each
        lane's d_label / d_filled.  Returns the number of rooms grown."""
        if len(self.growers) == 1 and self.growers[0].free_run and self.growers[0].n_rooms:
            # one free-running lane: the grower's own loop, whose rooms wait in the device-side queue and are taken by whichever slot
            # finishes (binding them here, between launches, would leave a finished slot idle for the rest of its launch and the
            # read-back pipeline behind it)
            with torch.cuda.stream(self.streams[0]):
                n = self.growers[0].grow_loaded(fill)
            torch.cuda.synchronize()
            return n
        queues, finished = [], 0
        for k, gr in enumerate(self.growers):
            with torch.cuda.stream(self.streams[k]):
                q = list(range(gr.n_rooms))
                if gr.n_rooms:
                    gr.reset_state()
                    for g in range(gr.n_groups):
                        gr.bind(g, q.pop(0) if q else -1)
                queues.append(q)
        total = sum(gr.n_rooms for gr in self.growers)
        live = [gr.n_rooms > 0 for gr in self.growers]
        done = [0] * len(self.growers)
        while finished < total:
            for k, gr in enumerate(self.growers):
                if not live[k]:
                    continue
                with torch.cuda.stream(self.streams[k]):
                    gr.enqueue()
                    gs = gr.poll_done()
                    if fill:
                        gr.fill_many([gr.group_room[g] for g in gs])
                    for g in gs:
                        finished += 1
                        done[k] += 1
                        gr.bind(g, queues[k].pop(0) if queues[k] else -1)
                    live[k] = done[k] < gr.n_rooms
        torch.cuda.synchronize()
        return total

    def collect(self, fill=True):
        per_lane = []
        for k, gr in enumerate(self.growers):
            with torch.cuda.stream(self.streams[k]):
                per_lane.append(gr.collect(fill) if gr.n_rooms else [])
        return [per_lane[k][j] for k, j in self.where]

    @property
    def instance_steps(self):
        return sum(gr.instance_steps for gr in self.growers)
