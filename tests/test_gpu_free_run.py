"""GPU: free-running launches (lrg_grow_async: every slot at its own pace inside one launch, csrc/lrg_async.inl) against the
lock-step iterations (lrg_grow_step_packed) and the oracle.  Same front code, same tile code on the same rows, so regions and
labels must agree exactly -- whatever the number of front workgroups, tile teams, steps per launch or the order in which the
slots happen to be served (test_region_grow.py:208-306 per slot; rooms are independent, :110-183).  (With nine slots the lock-step
launches sum the heads' pooled product on the matrix cores, the free-running kernel with vector FMAs in the same order.)"""
import numpy as np
import pytest

from conftest import seed_without_near_tie
from learn_region_grow_amd import synthetic
from oracle import grow_ref, rng_ref
from test_gpu_grow import WEIGHT_KW, SAME_LOGITS_MARGIN, gpu_net_fn, small_room, same_regions

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def net(cuda_device):
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    return LrgNetHIP(1, 1, 512, 512, 13, 0, device=cuda_device).load_weights(synthetic.make_synthetic_weights(**WEIGHT_KW))


def _rooms():
    return [small_room(400 + i, 600 + 200 * i, room_id=10 + i) for i in range(3)] + \
           [small_room(300, 1500, furniture=4, room_id=13), small_room(301, 2500, furniture=6, room_id=14)]


@pytest.mark.parametrize('steps,fronts,teams,in_flight,units', [(1, 0, 0, 5, 0), (7, 0, 0, 5, 0), (64, 1, 1, 3, 0), (64, 5, 3, 5, 0), (16, 2, 2, 2, 0), (64, 0, 0, 9, 0),
                                                                 (64, 3, 1, 5, -1), (16, 2, 2, 4, -1),
                                                                 (64, 0, 0, 30, 0), (64, 0, 0, 100, 0), (64, 0, 0, 200, 0),
                                                                 (64, 0, 4, 9, 0), (64, 0, 4, 9, -1), (64, 0, 0, 140, 0), (64, 0, 0, 260, 0)])
def test_free_run_equals_lock_step(net, steps, fronts, teams, in_flight, units):
    """units: 0 = the pooled-product units (the heads' pooled kernels in the LDS of workgroups of their own, head tiles started beside them),
    -1 = the pooled product as 128-column blocks of the tile teams.  in_flight 5 / 30 / 100 / 200: the shapes a launch takes by its
    slot count (one front workgroup per slot and one team per CU | units and two teams | units and three teams | no units, three teams); 140 / 260:
    units and four teams | no units and four teams (two branch-only teams on the smaller LDS region), also forced at 9 slots."""
    from learn_region_grow_amd.grow import RegionGrower
    rooms = _rooms()
    kw = dict(rooms_in_flight=in_flight, rng='counter', seed=123, policy='net')
    want = RegionGrower(net, free_run=False, **kw).run(rooms)
    # (free_run_waves=-1: the one-kernel launch whose tile teams run the branch tiles; the wave-branch launches have a test of their own below)
    gr = RegionGrower(net, free_run=True, free_run_steps=steps, free_run_fronts=fronts, free_run_teams=teams, free_run_units=units, free_run_waves=-1, **kw)
    got = gr.run(rooms)
    assert gr.free_run
    for g, w in zip(got, want):
        same_regions(g.regions, w.regions)
        np.testing.assert_array_equal(g.cluster_label, w.cluster_label)
        np.testing.assert_array_equal(g.filled_label, w.filled_label)


@pytest.mark.parametrize('steps,fronts,in_flight,units,waves', [(64, 0, 5, 0, 4), (1, 0, 5, 0, 4), (7, 2, 5, 0, 4), (64, 0, 9, 0, 8), (64, 0, 30, 0, 4), (64, 0, 68, 0, 4),
                                                                 (64, 3, 5, -1, 4), (16, 0, 100, -1, 4), (64, 0, 140, 0, 8), (64, 0, 200, 0, 4),
                                                                 (64, 0, 5, 0, 1), (1, 0, 5, 0, 1), (7, 2, 9, -1, 1), (64, 0, 30, 0, 1), (64, 0, 68, 0, 1), (16, 0, 200, 0, 1)])
def test_wave_branch_launches_equal_lock_step(net, steps, fronts, in_flight, units, waves):
    """Round 6 (csrc/lrg_wave_tile.inl): the launch as two kernels resident together -- front workgroups and units | the CUs that run tiles -- and a branch tile
    (learn_region_grow_util.py:106-123) as four one-wavefront tasks with the activations in registers, on CUs that hold the kernels of their (side, quarter) in LDS.
    The same sums in the same order as the team tiles: regions and labels equal the lock-step iterations' exactly.  waves: 4 = four wavefronts per wave-branch CU and a
    fill-in team beside them, 8 = eight (the fill-in between launches); 1 = REGISTER TILES: the same two kernels, a branch tile by a team of four wavefronts that keep
    layers 0 - 2 in registers and meet once (lrg_team_branch_tile_reg).  An option (LrgAsyncBuffers.branch_waves), off by default: measured slower than the one-kernel
    launch at every slot count."""
    from learn_region_grow_amd.grow import RegionGrower
    rooms = _rooms()
    kw = dict(rooms_in_flight=in_flight, rng='counter', seed=123, policy='net')
    want = RegionGrower(net, free_run=False, **kw).run(rooms)
    gr = RegionGrower(net, free_run=True, free_run_steps=steps, free_run_fronts=fronts, free_run_units=units, free_run_waves=waves, **kw)
    got = gr.run(rooms)
    assert gr.free_run
    for g, w in zip(got, want):
        same_regions(g.regions, w.regions)
        np.testing.assert_array_equal(g.cluster_label, w.cluster_label)
        np.testing.assert_array_equal(g.filled_label, w.filled_label)


def test_wave_branch_launch_that_cannot_be_resident_gives_up_cleanly(net, monkeypatch):
    """The two kernels of a wave-branch launch wait for each other, so all their workgroups must run at once.  With 64 worker workgroups more than the shader
    engines hold (a test hook) the launch can never be resident as a whole: its front workgroups give up at the start rendezvous within start_wait_us (reason 6),
    the late workgroups find the abort word and leave, the host raises instead of returning labels -- and the next launch on the same device is unharmed."""
    import time
    from learn_region_grow_amd import _lib
    from learn_region_grow_amd.grow import RegionGrower
    rooms = _rooms()
    kw = dict(rooms_in_flight=5, rng='counter', seed=123, policy='net')
    want = RegionGrower(net, free_run=False, **kw).run(rooms)
    monkeypatch.setenv('LRG_ASYNC_WAVE_EXTRA_WGS', '64')
    gr = RegionGrower(net, free_run=True, free_run_waves=4, **kw)
    gr.load_rooms(rooms)
    gr.async_buffers.start_wait_us = 30000
    t0 = time.time()
    with pytest.raises(_lib.LrgHipError, match='gave up'):
        gr.grow_loaded()
    assert time.time() - t0 < 20.0                      # (bounded by the rendezvous, not by the hand-overs' multi-second limits)
    monkeypatch.delenv('LRG_ASYNC_WAVE_EXTRA_WGS')
    got = RegionGrower(net, free_run=True, free_run_waves=4, **kw).run(rooms)
    for g, w in zip(got, want):
        same_regions(g.regions, w.regions)
        np.testing.assert_array_equal(g.filled_label, w.filled_label)


def test_two_kernel_launches_from_any_stream(net, cuda_device):
    """HIP streams share a few hardware queues, round robin by creation, and kernels of two streams on one queue never run side by side: the side stream of a
    two-kernel launch must not share the caller's queue (it is created at the highest priority: another pool).  With a side stream of default priority every fourth
    new caller stream made the worker kernel wait out its 4 s for a front kernel queued behind it.  Nine growers on nine fresh streams: all finish, with the labels."""
    import time
    import torch
    from learn_region_grow_amd.grow import RegionGrower
    rooms = _rooms()[:3]
    kw = dict(rooms_in_flight=3, rng='counter', seed=123, policy='net')
    want = RegionGrower(net, free_run=False, **kw).run(rooms)
    t0 = time.time()
    for k in range(9):
        with torch.cuda.stream(torch.cuda.Stream(device=cuda_device)):
            got = RegionGrower(net, free_run=True, free_run_waves=1, **kw).run(rooms)
            torch.cuda.synchronize()
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g.filled_label, w.filled_label)
    assert time.time() - t0 < 30.0


def test_free_run_matches_oracle(net):
    from learn_region_grow_amd.grow import RegionGrower
    rooms = _rooms()[:4]

    def oracle(seed):
        return [grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.CounterStream(seed, room['room_id']),
                                   net_fn=gpu_net_fn(net)) for room in rooms]
    seed, wants = seed_without_near_tie(oracle, range(123, 131), SAME_LOGITS_MARGIN)
    res = RegionGrower(net, rooms_in_flight=4, rng='counter', seed=seed, free_run=True, free_run_steps=32).run(rooms)
    for i, want in enumerate(wants):
        same_regions(res[i].regions, want.regions)
        np.testing.assert_array_equal(res[i].cluster_label, want.cluster_label)
        np.testing.assert_array_equal(res[i].filled_label, want.filled_label)


def test_free_run_alternates_with_lock_step(net):
    """A launch ends every slot between two evaluations, logits in place: free-running launches and lock-step iterations may
    alternate on the same buffers."""
    import torch
    from learn_region_grow_amd.grow import RegionGrower
    rooms = _rooms()[:3]
    kw = dict(rooms_in_flight=3, rng='counter', seed=7, policy='net')
    want = RegionGrower(net, free_run=False, **kw).run(rooms)
    gr = RegionGrower(net, free_run=True, free_run_steps=5, **kw)
    gr.load_rooms(rooms)
    for g in range(3):
        gr.bind(g, g)
    for k in range(100000):
        if k % 2:
            gr.enqueue_free_run()
        else:
            for _ in range(3):
                gr.enqueue_iteration()
        if k % 8 == 7:
            torch.cuda.synchronize()
            if int(gr.d_stats[1].item()) >= 3:
                break
    for r in range(3):
        gr.fill(r)
    got = gr.collect()
    for g, w in zip(got, want):
        same_regions(g.regions, w.regions)
        np.testing.assert_array_equal(g.filled_label, w.filled_label)


def test_free_run_refuses_a_stream_with_too_few_compute_units(net, cuda_device):
    """A free-running launch is one workgroup per CU whose roles wait for each other: on a stream confined to fewer CUs than the launch has
    workgroups (hipExtStreamCreateWithCUMask) it could never be resident as a whole.  Refused with LRG_ERESIDENCY before anything is
    enqueued -- not found out by a spin bound seconds later (the reference has one session and one device to itself, test_region_grow.py:86-90).
    Told how many CUs the stream may use (LrgAsyncBuffers.compute_units), the same launch runs there and gives the lock-step labels."""
    import ctypes
    import time
    import torch
    from learn_region_grow_amd import _lib
    from learn_region_grow_amd.grow import RegionGrower
    lib = _lib.load()
    ncu = torch.cuda.get_device_properties(cuda_device).multi_processor_count
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    allowed = ncu // 2
    for b in range(allowed):
        mask[b // 32] |= 1 << (b % 32)
    h = ctypes.c_void_p()
    _lib.check(lib.lrg_stream_create_cu_mask(mask, words, ctypes.byref(h)), 'lrg_stream_create_cu_mask')
    stream = torch.cuda.ExternalStream(h.value, device=cuda_device)
    rooms = _rooms()[:3]
    want = RegionGrower(net, rooms_in_flight=3, rng='counter', seed=5, free_run=False).run(rooms)
    try:
        with torch.cuda.stream(stream):
            gr = RegionGrower(net, rooms_in_flight=3, rng='counter', seed=5, free_run=True)
            gr.load_rooms(rooms)
            gr.free_run_begin()
            t0 = time.perf_counter()
            rc = lib.lrg_grow_async(gr.d_slots.data_ptr(), gr.d_rooms.data_ptr(), gr.S, gr.cap, ctypes.byref(gr.params), ctypes.byref(gr.net._w),
                                    ctypes.byref(gr.packed_buffers), ctypes.byref(gr.async_buffers), 16, 1000, ctypes.c_void_p(stream.cuda_stream))
            assert rc == _lib.LRG_ERESIDENCY and time.perf_counter() - t0 < 0.5
            with pytest.raises(_lib.LrgHipError, match='LRG_ERESIDENCY'):
                gr.enqueue_free_run()
            stream.synchronize()
            assert int(gr.d_stats[3].item()) == 0 and int(gr.d_stats[2].item()) == 0      # nothing ran, nothing gave up
            # the same stream, the launch sized for it
            gr.async_buffers.compute_units = allowed
            got = gr._grow_loaded_free_run(True)
            res = gr.collect(True)
        for g, w in zip(res, want):
            same_regions(g.regions, w.regions)
            np.testing.assert_array_equal(g.filled_label, w.filled_label)
    finally:
        torch.cuda.synchronize()      # (the stream is left to the process, as grow.fill_streams leaves its masked streams: events recorded on it outlive the test)


def test_fill_in_inside_the_launch_equals_the_fill_in_between_launches(net, monkeypatch):
    """The rooms that finish during a free-running launch get their 1-NN fill-in (test_region_grow.py:308-316) from tile teams of the same
    launch (LrgAsyncBuffers.fill_list: one task per 256 candidate points, the last one writes the filled labels) -- the done ring says so
    (bit 31 of the slot word), the host launches nothing, and the filled labels are those of lrg_nn1_fill_batch and of the C oracle."""
    from learn_region_grow_amd.grow import RegionGrower
    from oracle import grouping_ref
    rooms = _rooms() + [small_room(310, 4000, furniture=8, room_id=15)]
    kw = dict(rooms_in_flight=3, rng='counter', seed=11, policy='net')
    # (short launches: rooms finish in many different ones; the one-kernel launch: a register-tile launch leaves the fill-ins to the host)
    gr = RegionGrower(net, free_run=True, free_run_budget_us=300, free_run_waves=-1, **kw)
    calls = []
    orig = gr.fill_many
    gr.fill_many = lambda rs: (calls.append(list(rs)), orig(rs))[1]
    got = gr.run(rooms)
    assert gr.free_run and gr.fill_in_launch and sum(len(c) for c in calls) == 0      # nothing left for the host to fill in
    monkeypatch.setenv('LRG_FREE_RUN_FILL', '0')
    ref = RegionGrower(net, free_run=True, **kw)
    want = ref.run(rooms)
    assert not ref.fill_in_launch
    for room, g, w in zip(rooms, got, want):
        same_regions(g.regions, w.regions)
        np.testing.assert_array_equal(g.cluster_label, w.cluster_label)
        np.testing.assert_array_equal(g.filled_label, w.filled_label)
        np.testing.assert_array_equal(g.filled_label, grouping_ref.nn1_fill(room['points'], g.cluster_label))


@pytest.mark.parametrize('F,lite', [(12, 0), (9, 0), (13, 2)])
def test_free_run_feature_size_and_lite_variants(cuda_device, F, lite):
    """The reference's feature-size variants (test_region_grow.py:72-77: the first F of the 13 columns) and lite = 2 through the free-running launches --
    rows gathered at a 64-byte stride in 16-byte pieces whatever F is (9 .. 16), the host's fill-in kernels where the in-launch one (13 features) does
    not apply -- against the lock-step iterations."""
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    from learn_region_grow_amd.grow import RegionGrower
    w = synthetic.make_synthetic_weights(feature_size=F, lite=lite, **WEIGHT_KW)
    net = LrgNetHIP(1, 1, 512, 512, F, lite, device=cuda_device).load_weights(w)
    rooms = [dict(r, points=np.ascontiguousarray(r['points'][:, :F])) for r in _rooms()[:4]]
    kw = dict(rooms_in_flight=4, rng='counter', seed=3, policy='net')
    want = RegionGrower(net, free_run=False, **kw).run(rooms)
    gr = RegionGrower(net, free_run=True, **kw)
    got = gr.run(rooms)
    assert gr.free_run and gr.fill_in_launch == (F == 13)
    for g, w_ in zip(got, want):
        same_regions(g.regions, w_.regions)
        np.testing.assert_array_equal(g.cluster_label, w_.cluster_label)
        np.testing.assert_array_equal(g.filled_label, w_.filled_label)


@pytest.mark.parametrize('in_flight,tail_rows,close_us,units,teams', [(5, 4096, 0, 0, 0), (9, 4096, -1, 0, 0), (9, 256, 0, 0, 0), (30, 8192, 30, 0, 0), (100, 16384, 0, 0, 0),
                                                                       (200, 32768, 0, 0, 0), (30, 8192, 0, -1, 0), (9, 4096, 0, 0, 4), (64, 64, 0, 0, 0)])
def test_shared_tail_tiles_equal_lock_step(net, monkeypatch, in_flight, tail_rows, close_us, units, teams):
    """Shared tail tiles (LrgAsyncBuffers.tail_ctl): the rows beyond a slot's last full 32-row tile share branch tiles with other slots' tails (the packed tile
    form: runs of rows, per-run max-pool), the head stack of a tail is a tile of the slot's own that stores only the slot's rows.  Same regions and labels as the
    lock-step iterations -- with tiles that fill up by themselves (many slots), tiles closed at once (close_us -1) or after a long wait (30 us), 256 / 64 shared
    rows (the launches run out of them and fall back to padded tiles of the slots' own), the pooled product by the units or as blocks of the tile teams, four teams."""
    from learn_region_grow_amd.grow import RegionGrower
    rooms = _rooms()
    kw = dict(rooms_in_flight=in_flight, rng='counter', seed=123, policy='net')
    want = RegionGrower(net, free_run=False, **kw).run(rooms)
    monkeypatch.setenv('LRG_FREE_RUN_TAIL_US', str(close_us))
    gr = RegionGrower(net, free_run=True, free_run_tail_rows=tail_rows, free_run_units=units, free_run_teams=teams, **kw)
    got = gr.run(rooms)
    assert gr.free_run and gr.tail_rows == tail_rows and gr.async_buffers.tail_ctl
    for g, w in zip(got, want):
        same_regions(g.regions, w.regions)
        np.testing.assert_array_equal(g.cluster_label, w.cluster_label)
        np.testing.assert_array_equal(g.filled_label, w.filled_label)
    work = gr.a_work.cpu().numpy()
    assert work[3] * 32 >= work[1] + work[2]          # (tiles run x 32 rows cover the rows evaluated)


def test_shared_tail_tiles_across_short_launches(net):
    """Every launch hands the shared rows out again from row 0, and a slot enters a launch with the logits of its last evaluation still to be read: they are
    copied to the slot's own rows before it leaves a launch (lrg_async_tail_logits_home), so that the next launch's tails cannot land on them.  Provoked here with
    launches of ONE evaluation per slot that alternate between 64 front workgroups (the tails reserved in any order) and 4 (sixteen slots each: a workgroup's last
    slots take their first turn ~300 us into the launch, when the tails of 40 evaluations have been written from row 0 on): with the hazard compiled back in
    (-DLRG_EXP_NO_TAIL_HOME) rooms come out with other labels; as built, regions and labels are the lock-step iterations'."""
    import torch
    from learn_region_grow_amd.grow import RegionGrower
    rooms = [small_room(700 + i, 420 + 37 * (i % 9), room_id=40 + i) for i in range(64)]
    kw = dict(rooms_in_flight=64, rng='counter', seed=5, policy='net')
    want = RegionGrower(net, free_run=False, **kw).run(rooms)
    gr = RegionGrower(net, free_run=True, free_run_steps=1, free_run_tail_rows=2048, free_run_waves=-1, **kw)
    gr.load_rooms(rooms)
    assert gr.free_run and gr.async_buffers.tail_ctl
    for g in range(64):
        gr.bind(g, g)
    for k in range(100000):
        gr.async_buffers.front_workgroups = 64 if k % 2 == 0 else 4
        gr.enqueue_free_run()
        if k % 16 == 15:
            torch.cuda.synchronize()
            assert int(gr.d_stats[3].item()) == 0
            if int(gr.d_stats[1].item()) >= 64:
                break
    for r in range(64):
        gr.fill(r)
    got = gr.collect()
    for g, w in zip(got, want):
        same_regions(g.regions, w.regions)
        np.testing.assert_array_equal(g.cluster_label, w.cluster_label)
        np.testing.assert_array_equal(g.filled_label, w.filled_label)


def test_shared_tail_tiles_with_speculation(net, monkeypatch):
    from learn_region_grow_amd.grow import RegionGrower
    rooms = _rooms()
    kw = dict(rooms_in_flight=3, rng='counter', seed=123, policy='net')
    want = RegionGrower(net, free_run=True, **kw).run(rooms)
    got = RegionGrower(net, speculate=3, free_run_tail_rows=4096, **kw).run(rooms)
    for g, w in zip(got, want):
        same_regions(g.regions, w.regions)
        np.testing.assert_array_equal(g.filled_label, w.filled_label)


@pytest.mark.parametrize('in_flight,patience_us,teams,tail_rows', [(5, 1.5, 0, 0), (9, 0.1, 0, 0), (30, 5, 0, 0), (100, 1.5, 0, 0), (200, 1.5, 0, 0), (260, 1.5, 0, 32768), (9, 1.5, 4, 4096)])
def test_batched_pooled_products_equal_lock_step(net, monkeypatch, in_flight, patience_us, teams, tail_rows):
    """Without the pooled-product units the slots whose branch tiles are in queue up and the heads' pooled products are computed for up to eight of them per trip of
    the kernels' columns from L2 (lrg_async.inl, LRG_GEMV_BATCH) -- the same sums in the same order as one slot at a time: same regions and labels as the lock-step
    iterations, with batches that fill up (many slots), batches closed after 0.1 us (mostly single slots) or 5 us, with shared tail tiles, with four teams."""
    from learn_region_grow_amd.grow import RegionGrower
    rooms = _rooms()
    kw = dict(rooms_in_flight=in_flight, rng='counter', seed=123, policy='net')
    want = RegionGrower(net, free_run=False, **kw).run(rooms)
    monkeypatch.setenv('LRG_ASYNC_GEMV_BATCH', '1')
    monkeypatch.setenv('LRG_ASYNC_GEMV_BATCH_US', str(patience_us))
    gr = RegionGrower(net, free_run=True, free_run_units=-1, free_run_teams=teams, free_run_tail_rows=tail_rows, **kw)
    got = gr.run(rooms)
    assert gr.free_run
    for g, w in zip(got, want):
        same_regions(g.regions, w.regions)
        np.testing.assert_array_equal(g.cluster_label, w.cluster_label)
        np.testing.assert_array_equal(g.filled_label, w.filled_label)


def test_room_order_does_not_change_the_labels(net):
    """More rooms than two rounds of slots: they take slots in dist.queue_order (long rooms early, sizes mixed) instead of as loaded -- rooms are independent
    (test_region_grow.py:110-183) and the random stream is keyed by the room, so regions and labels are those of the loaded order and of the lock-step iterations;
    with four teams per worker CU forced, the fill-in teams take ring 1's tasks between fill-ins here too."""
    from learn_region_grow_amd.grow import RegionGrower
    base = _rooms()
    rooms = [dict(base[j % len(base)], room_id=500 + j) for j in range(23)]
    kw = dict(rooms_in_flight=4, rng='counter', seed=7, policy='net')
    want = RegionGrower(net, free_run=False, room_order='loaded', **kw).run(rooms)
    made = []
    for order, free, teams in (('queue', True, 0), ('loaded', True, 0), ('queue', False, 0), ('queue', True, 4)):
        gr = RegionGrower(net, free_run=free, room_order=order, free_run_budget_us=1500, free_run_teams=teams, **kw)
        got = gr.run(rooms)
        made.append(gr)
        for a, b in zip(want, got):
            same_regions(a.regions, b.regions)
            np.testing.assert_array_equal(a.cluster_label, b.cluster_label)
            np.testing.assert_array_equal(a.filled_label, b.filled_label)
    assert made[0].room_order() != list(range(len(rooms))) and sorted(made[0].room_order()) == list(range(len(rooms)))
    assert made[1].room_order() == list(range(len(rooms)))
