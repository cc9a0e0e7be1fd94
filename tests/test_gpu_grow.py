"""GPU: the batched region-grow loop (HIP kernels through the C-ABI) against the oracle and the goldens made
by running the reference's own scripts.

Decomposition: the network is checked to tolerance in test_gpu_net.py; here the oracle loop is
driven with the SAME GPU network (net_fn) so that every integer / index / mask result must agree exactly."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, seed_without_near_tie
from learn_region_grow_amd import synthetic, preprocess
from oracle import grow_ref, rng_ref

pytestmark = pytest.mark.gpu
WEIGHT_KW = dict(seed=0, gain=2.0, bias_std=0.2, add_bias_shift=0.0, rmv_bias_shift=-3.0)
# Bernoulli draws are compared as u < conf.  With the SAME logits on both sides only exp/divide rounding differs
# (~1e-7 relative); against an independent fp32 evaluation of the network the logits differ by ~1e-5 relative.
SAME_LOGITS_MARGIN = 5e-7
OTHER_NETWORK_MARGIN = 1e-3


@pytest.fixture(scope='module')
def net(cuda_device):
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    return LrgNetHIP(1, 1, 512, 512, 13, 0, device=cuda_device).load_weights(synthetic.make_synthetic_weights(**WEIGHT_KW))


@pytest.fixture(scope='module')
def net_trained(cuda_device):
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    return LrgNetHIP(1, 1, 512, 512, 13, 0, device=cuda_device).load_weights(synthetic.load_trained_weights())


def gpu_net_fn(net):
    def fn(xi, xn):
        _, add, _, rmv, _ = net.run(xi, xn)
        return add, rmv
    return fn


def golden_room(name, room_id=0):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    return dict(points=g['points'], obj_id=g['obj_id'], order=g['order'], room_id=room_id), g


def small_room(seed, n_raw, furniture=0, room_id=0):
    raw = (synthetic.area5_shaped_room(n_raw, seed, n_furniture=furniture) if furniture
           else synthetic.generate_room_points(n_raw, seed)).astype(np.float32)
    p = preprocess.preprocess_room(raw[:, :6], raw[:, 6].astype(int), raw[:, 7].astype(int))
    return dict(points=p['points'], obj_id=p['obj_id'], order=p['order'], room_id=room_id)


def same_regions(got, want):
    g = [(r['seed'], r['steps'], r['points'], r['reason'], r['labeled']) for r in got]
    w = [(r['seed'], r['steps'], r['points'], r['reason'], r['labeled']) for r in want]
    assert g == w, 'first difference at region %d: %s vs %s' % (
        next(i for i, (a, b) in enumerate(zip(g + [None], w + [None])) if a != b),
        (g + [None])[next(i for i, (a, b) in enumerate(zip(g + [None], w + [None])) if a != b)],
        (w + [None])[next(i for i, (a, b) in enumerate(zip(g + [None], w + [None])) if a != b)])


def test_legacy_single_room_step_by_step(net):
    """One room, reference-order RNG: every step's sampled sets, centre, stacked inputs, masks and the mask
    itself equal the oracle's; final labels equal the oracle's and the reference script's own output."""
    from learn_region_grow_amd.grow import RegionGrower
    room, g = golden_room('greedy_room101')
    want_steps = []
    want = grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.LegacyStream(0),
                              net_fn=gpu_net_fn(net), hook=want_steps.append)
    got_steps = []
    gr = RegionGrower(net, rooms_in_flight=1, rng='legacy', advance_rounds=1)
    gr.debug_hook = got_steps.append
    res = gr.run([room])[0]
    n = min(len(got_steps), len(want_steps))
    for i in range(n):
        a, b = got_steps[i], want_steps[i]
        ctx = 'step %d (seed %d)' % (i, b['seed'])
        assert (a['seed'], a['step'], a['nc'], a['ne']) == (b['seed'], b['step'], b['nc'], b['ne']), ctx
        np.testing.assert_array_equal(a['mask_before'], b['mask_before'], err_msg=ctx)
        np.testing.assert_array_equal(a['min_dims'], b['min_dims'], err_msg=ctx)
        np.testing.assert_array_equal(a['subset_in'], b['subset_in'], err_msg=ctx)
        cen = b['center'].copy()
        cen[2:6] = 0
        np.testing.assert_array_equal(a['center'][:13], cen, err_msg=ctx)
        np.testing.assert_array_equal(a['inlier'], b['inlier'][0], err_msg=ctx)
        np.testing.assert_array_equal(a['neighbor'], b['neighbor'][0], err_msg=ctx)
        np.testing.assert_array_equal(a['add'], b['add'][0], err_msg=ctx)
        np.testing.assert_array_equal(a['add_mask'], b['add_mask'], err_msg=ctx)
        np.testing.assert_array_equal(a['rmv_mask'], b['rmv_mask'], err_msg=ctx)
    assert len(got_steps) == len(want_steps)
    same_regions(res.regions, want.regions)
    np.testing.assert_array_equal(res.cluster_label, want.cluster_label)
    np.testing.assert_array_equal(res.filled_label, want.filled_label)
    # against the reference script's output: identical unless a Bernoulli draw sits within fp32 noise of its
    # confidence (oracle margin for this room: 1.4e-5)
    np.testing.assert_array_equal(res.filled_label, g['filled_label'])


@pytest.mark.parametrize('name,restarts', [('greedy_room100', 1), ('restart_room103', 10)])
def test_legacy_matches_reference_script_output(net, name, restarts):
    from learn_region_grow_amd.grow import RegionGrower
    room, g = golden_room(name)
    want = grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.LegacyStream(0),
                              net_fn=gpu_net_fn(net), restarts=0 if restarts == 1 else restarts)
    res = RegionGrower(net, rooms_in_flight=1, rng='legacy', restarts=restarts).run([room])[0]
    same_regions(res.regions, want.regions)
    np.testing.assert_array_equal(res.filled_label, want.filled_label)
    # against the reference script's own output (tests/golden: test_region_grow.py / test_random_restart.py run unmodified): the inputs are
    # fixed, so this either holds or it does not -- the closest Bernoulli draw of these two runs keeps a relative distance of
    # > 1e-5 from its confidence (the oracle's min_rel_margin), far above what float32 rounding of the logits can move
    assert want.min_rel_margin > OTHER_NETWORK_MARGIN / 100
    np.testing.assert_array_equal(res.filled_label, g['filled_label'])


@pytest.mark.parametrize('name,restarts,labeled', [('greedy_trained_room114', 1, 51), ('restart_trained_room137', 10, 20)])
def test_legacy_matches_reference_script_output_under_trained_weights(net_trained, name, restarts, labeled):
    """The reference's scripts run unmodified under the weights this repository trained (tests/golden/make_golden.py trained): realistic
    dynamics -- 51 labelled regions over 518 network steps (test_region_grow.py:208-316), and 20 labelled regions of ten restarts each of
    which 15 are won by a restart other than the first (argmax(restart_score), test_random_restart.py:177).  Labels must equal the
    script's own output; regions and labels must equal the oracle loop driven by the GPU network."""
    from learn_region_grow_amd.grow import RegionGrower
    room, g = golden_room(name)
    want = grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.LegacyStream(0),
                              net_fn=gpu_net_fn(net_trained), restarts=0 if restarts == 1 else restarts)
    res = RegionGrower(net_trained, rooms_in_flight=1, rng='legacy', restarts=restarts).run([room])[0]
    same_regions(res.regions, want.regions)
    np.testing.assert_array_equal(res.filled_label, want.filled_label)
    lab = [r for r in want.regions if r['labeled']]
    assert len(lab) == labeled and {r['reason'] for r in lab} == {'noexpand', 'stuck', 'noneighbor'}
    if restarts > 1:
        assert sum(1 for r in lab if r['best_restart'] != 0) >= 10
    # the rooms were picked (of twelve each) for the distance their closest Bernoulli draw keeps from its confidence in the NumPy
    # evaluation of the network (1.8e-4 / 2.0e-5 relative): float32 rounding differences between the GPU's and NumPy's logits do not
    # reach it, so the script's own labels are reproduced exactly
    assert want.min_rel_margin > OTHER_NETWORK_MARGIN / 100
    np.testing.assert_array_equal(res.filled_label, g['filled_label'])


def test_legacy_many_rooms_in_flight(net):
    """Rooms are independent: 5 rooms through 2 slots (queue + refill) equal 5 separate oracle runs."""
    from learn_region_grow_amd.grow import RegionGrower
    rooms = [small_room(200 + i, 500 + 150 * i, room_id=i) for i in range(4)] + [small_room(300, 1500, furniture=4, room_id=4)]
    res = RegionGrower(net, rooms_in_flight=2, rng='legacy').run(rooms)
    for i, room in enumerate(rooms):
        want = grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.LegacyStream(i),
                                  net_fn=gpu_net_fn(net))
        same_regions(res[i].regions, want.regions)
        np.testing.assert_array_equal(res[i].filled_label, want.filled_label)


@pytest.mark.parametrize('restarts,group', [(1, 1), (4, 4), (5, 2)])
def test_counter_stream_matches_oracle(net, restarts, group):
    """Device-side randomness (Philox counter stream), restarts batched over slot groups."""
    from learn_region_grow_amd.grow import RegionGrower
    rooms = [small_room(400 + i, 600 + 200 * i, room_id=10 + i) for i in range(3)]

    def oracle(seed):
        return [grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None,
                                   rng_ref.CounterStream(seed, room['room_id']), net_fn=gpu_net_fn(net),
                                   restarts=0 if restarts == 1 else restarts) for room in rooms]
    seed, wants = seed_without_near_tie(oracle, range(123, 131), SAME_LOGITS_MARGIN)
    res = RegionGrower(net, rooms_in_flight=2, rng='counter', seed=seed, restarts=restarts, group_size=group).run(rooms)
    for i, want in enumerate(wants):
        same_regions(res[i].regions, want.regions)
        np.testing.assert_array_equal(res[i].cluster_label, want.cluster_label)
        np.testing.assert_array_equal(res[i].filled_label, want.filled_label)


@pytest.mark.parametrize('policy', ['gt', 'threshold'])
def test_policies_and_step_cap(net, policy):
    from learn_region_grow_amd.grow import RegionGrower
    room = small_room(500, 1200, furniture=3, room_id=7)
    res = RegionGrower(net, rooms_in_flight=1, rng='counter', seed=5, policy=policy, max_region_steps=20).run([room])[0]
    want = grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.CounterStream(5, 7),
                              net_fn=gpu_net_fn(net), policy=policy, max_region_steps=20)
    same_regions(res.regions, want.regions)
    np.testing.assert_array_equal(res.filled_label, want.filled_label)


def test_results_do_not_depend_on_batching(net):
    """Counter-stream draws are a function of (room, seed point, restart, step): 1 or 3 rooms in flight, any order."""
    from learn_region_grow_amd.grow import RegionGrower
    rooms = [small_room(600 + i, 700, room_id=20 + i) for i in range(3)]
    a = RegionGrower(net, rooms_in_flight=1, rng='counter', seed=9).run(rooms)
    b = RegionGrower(net, rooms_in_flight=3, rng='counter', seed=9).run(rooms[::-1])[::-1]
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x.filled_label, y.filled_label)


@pytest.mark.parametrize('lanes,in_flight', [(2, 4), (3, 3), (4, 2)])
def test_lanes_do_not_change_results(net, lanes, in_flight):
    """Two or more lanes (half-batches on their own HIP streams, LanedRegionGrower) give the labels of one lane; more lanes
    than rooms in flight, or than rooms, degrade gracefully."""
    from learn_region_grow_amd.grow import LanedRegionGrower, RegionGrower
    rooms = [small_room(610 + i, 500 + 120 * i, room_id=40 + i) for i in range(5)]
    a = RegionGrower(net, rooms_in_flight=2, rng='counter', seed=11, policy='gt').run(rooms)
    b = LanedRegionGrower(net, rooms_in_flight=in_flight, lanes=lanes, rng='counter', seed=11, policy='gt').run(rooms)
    assert [x.room_id for x in b] == [x.room_id for x in a]
    for x, y in zip(a, b):
        same_regions(y.regions, x.regions)
        np.testing.assert_array_equal(x.cluster_label, y.cluster_label)
        np.testing.assert_array_equal(x.filled_label, y.filled_label)


@pytest.mark.parametrize('restarts', [1, 3])
def test_packed_iterations_equal_the_nine_launch_step(net, restarts):
    """lrg_grow_step_packed (one fused front kernel per slot + the network on packed distinct rows) against lrg_grow_step (the
    nine-launch chain on padded per-slot tiles), a HIP-graph replay of four packed iterations per host call, and (greedy growing)
    the free-running launches of lrg_grow_async: same regions, same labels.  Rooms from 0.5 k to 6 k points, so regions above
    512 / 1024 / 4096 points occur."""
    import torch
    from learn_region_grow_amd.grow import RegionGrower
    rooms = [small_room(620 + i, n, furniture=f, room_id=50 + i) for i, (n, f) in enumerate([(500, 0), (900, 2), (2500, 3), (6000, 4)])]
    kw = dict(rooms_in_flight=3, rng='counter', seed=13, restarts=restarts)
    for policy in ('gt', 'net'):
        a = RegionGrower(net, packed=False, policy=policy, **kw)
        ra = a.run(rooms)
        b = RegionGrower(net, packed=True, free_run=False, policy=policy, **kw)
        rb = b.run(rooms)
        assert not a.packed and b.packed and not b.free_run
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            c = RegionGrower(net, packed=True, free_run=False, graph_iterations=4, policy=policy, **kw)
            rc = c.run(rooms)
        assert c._graph is not None
        d = RegionGrower(net, packed=True, policy=policy, **kw)          # free-running launches where they apply (greedy growing)
        rd = d.run(rooms)
        assert d.free_run == (restarts == 1)
        for x, y, z, w in zip(ra, rb, rc, rd):
            same_regions(y.regions, x.regions)
            same_regions(z.regions, x.regions)
            same_regions(w.regions, x.regions)
            np.testing.assert_array_equal(x.cluster_label, y.cluster_label)
            np.testing.assert_array_equal(x.filled_label, y.filled_label)
            np.testing.assert_array_equal(x.filled_label, z.filled_label)
            np.testing.assert_array_equal(x.filled_label, w.filled_label)
            assert x.total_steps == y.total_steps == z.total_steps == w.total_steps


def test_unequalised_room_is_rejected(net):
    from learn_region_grow_amd.grow import RegionGrower
    from learn_region_grow_amd._lib import LrgHipError
    room = small_room(700, 500)
    room['points'] = np.concatenate([room['points'], room['points'][:1]])     # two points in one voxel
    room['obj_id'] = np.concatenate([room['obj_id'], room['obj_id'][:1]])
    room['order'] = np.arange(len(room['points']))
    with pytest.raises(LrgHipError):
        RegionGrower(net, rooms_in_flight=1).load_rooms([room])


@pytest.mark.parametrize('F,lite', [(9, 0), (6, 1), (12, 2)])
def test_feature_size_and_lite_variants(cuda_device, F, lite):
    """test_region_grow.py:72-77 feature-size variants keep the first F of the 13 columns; --lite picks the small
    networks.  The centred channel set (:243-247) changes with F."""
    from learn_region_grow_amd.grow import RegionGrower
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    w = synthetic.make_synthetic_weights(feature_size=F, lite=lite, **WEIGHT_KW)
    netv = LrgNetHIP(1, 1, 512, 512, F, lite, device=cuda_device).load_weights(w)
    room = small_room(800 + F, 900, furniture=3, room_id=F)
    room['points'] = np.ascontiguousarray(room['points'][:, :F])

    def oracle(seed):
        return [grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.CounterStream(seed, F),
                                   net_fn=gpu_net_fn(netv), lite=lite)]
    seed, (want,) = seed_without_near_tie(oracle, range(4, 12), SAME_LOGITS_MARGIN)
    res = RegionGrower(netv, rooms_in_flight=1, rng='counter', seed=seed).run([room])[0]
    same_regions(res.regions, want.regions)
    np.testing.assert_array_equal(res.filled_label, want.filled_label)


@pytest.mark.parametrize('n', [1, 5, 12])
def test_tiny_rooms(net, n):
    """Rooms smaller than the cluster threshold (:30): every region is dropped (:213) and the fill-in has nothing to
    copy from -- labels stay 0, as defined for the path (the reference would raise at :314)."""
    from learn_region_grow_amd.grow import RegionGrower
    room = small_room(900 + n, 600)
    keep = np.arange(n) * 7
    room = dict(points=np.ascontiguousarray(room['points'][keep]), obj_id=room['obj_id'][keep],
                order=np.arange(n)[::-1].copy(), room_id=n)
    res = RegionGrower(net, rooms_in_flight=1, rng='counter', seed=1).run([room])[0]
    want = grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.CounterStream(1, n),
                              net_fn=gpu_net_fn(net))
    same_regions(res.regions, want.regions)
    np.testing.assert_array_equal(res.filled_label, want.filled_label)
    assert sum(r['points'] for r in res.regions) == n


@pytest.mark.parametrize('pattern', ['scattered', 'corner', 'none', 'all'])
def test_room_fill_in(net, pattern):
    """RegionGrower.fill (tiled 1-NN search) equals the oracle's exhaustive search on every labelling: scattered holes,
    labels only in one corner, nothing labeled, everything labeled."""
    import torch
    from learn_region_grow_amd.grow import RegionGrower
    room = small_room(77, 2500, furniture=3, room_id=3)
    n = len(room['points'])
    rs = np.random.RandomState(1)
    if pattern == 'scattered':
        lab = ((rs.rand(n) < 0.7) * rs.randint(1, 30, n)).astype(np.int32)
    elif pattern == 'corner':
        lab = ((room['points'][:, 3] < 0.2) & (room['points'][:, 4] < 0.3)).astype(np.int32) * rs.randint(1, 5, n).astype(np.int32)
    elif pattern == 'none':
        lab = np.zeros(n, np.int32)
    else:
        lab = rs.randint(1, 9, n).astype(np.int32)
    gr = RegionGrower(net, rooms_in_flight=1, rng='counter')
    gr.load_rooms([room])
    gr.d_label[:n].copy_(torch.from_numpy(lab).to(gr.d_label.device))
    gr.fill(0)
    torch.cuda.synchronize()
    got = gr.d_filled[:n].cpu().numpy()
    want = grow_ref.fill_unlabeled(room['points'], lab.astype(np.int64)) if (lab != 0).any() else lab
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('with_chan_major', [True, False])
def test_medians_of_large_regions(net, monkeypatch, with_chan_major):
    """numpy.median (:241) over regions of every size class of lrg_median -- one wavefront, the radix select with 4 / 16 keys per
    thread, the sampled-pivot bisection above 16 Ki points, the LDS-cached selection above 48 Ki -- on channels that are smooth,
    sorted along the list, two-valued, constant, or constant but for a few outliers; odd and even counts."""
    import ctypes
    import torch
    from learn_region_grow_amd import _lib
    from learn_region_grow_amd.grow import RegionGrower
    from learn_region_grow_amd._lib import LrgSlot, LRG_ACTIVE
    from learn_region_grow_amd.lrgnet import _ptr, _stream_ptr
    if not with_chan_major:
        monkeypatch.setenv('LRG_NO_CHAN_MAJOR', '1')
    N = 70001
    rs = np.random.RandomState(7)
    pts = np.zeros((N, 13), dtype=np.float32)
    grid = np.stack(np.meshgrid(np.arange(420), np.arange(170), indexing='ij'), -1).reshape(-1, 2)[:N]
    pts[:, 0] = grid[:, 0] * 0.1 + 0.03
    pts[:, 1] = grid[:, 1] * 0.1 + 0.04                    # one point per 0.1 m voxel (the loader insists)
    pts[:, 2] = 0.05
    pts[:, 6] = rs.rand(N)                                 # smooth
    pts[:, 7] = np.sort(rs.randn(N)).astype(np.float32)    # sorted along the list
    pts[:, 8] = (rs.rand(N) < 0.5)                         # two values
    pts[:, 9] = 0.25                                       # constant
    pts[:, 10] = np.where(rs.rand(N) < 0.001, rs.randn(N), 1.0)      # constant but for a few outliers
    pts[:, 11] = -np.abs(rs.randn(N)) * 1e-3               # negative, clustered near zero
    pts[:, 12] = rs.randint(0, 50, N) * 0.5                # heavy duplicates
    room = dict(points=pts, obj_id=np.zeros(N, np.int32), order=np.arange(N, dtype=np.int32), room_id=0)
    sizes = [200, 257, 1000, 1025, 4000, 4097, 16000, 16385, 30000, 40001, 49152, 49153, 70000]
    gr = RegionGrower(net, rooms_in_flight=len(sizes), rng='counter', seed=1, policy='gt')
    gr.load_rooms([room])
    sz = ctypes.sizeof(LrgSlot)
    lists = []
    for s, nc in enumerate(sizes):
        idx = np.sort(rs.permutation(N)[:nc]).astype(np.int32) if s % 2 else rs.permutation(N)[:nc].astype(np.int32)
        lists.append(idx)
        gr.d_curidx[s, :nc] = torch.from_numpy(idx).to(gr.dev)
        sl = gr.h_slots[s]
        sl.room, sl.status, sl.nc = 0, LRG_ACTIVE, nc
    gr.d_slots.copy_(torch.from_numpy(np.frombuffer(bytes(gr.h_slots), dtype=np.uint8).copy()))
    center = torch.full((len(sizes), 16), -7.0, dtype=torch.float32, device=gr.dev)
    _lib.check(gr.lib.lrg_median(_ptr(gr.d_slots), _ptr(gr.d_rooms), len(sizes), ctypes.byref(gr.params), _ptr(center),
                                 _stream_ptr(gr.dev)), 'lrg_median')
    got = center.cpu().numpy()
    for s, idx in enumerate(lists):
        want = np.median(pts[idx], axis=0)
        for ch in (0, 1, 6, 7, 8, 9, 10, 11, 12):
            assert got[s, ch] == want[ch], 'region of %d points, channel %d: %r vs %r' % (len(idx), ch, got[s, ch], want[ch])
        assert (got[s, 2:6] == 0).all()
