"""GPU: BASELINE.json's named configurations, each checked on labels (not only timed):
  configs[1]  the benchmark configuration itself -- 68 Area-5-shaped rooms in flight over the lanes bench.py chooses (three);
  configs[2]  ScanNet-shaped rooms, 8 in flight;
  configs[3]  test_random_restart.py --scoring np with 16 restarts per seed batched per launch (1088 slots for the 68-room set).
Exact parity where the oracle finishes in seconds (ground-truth masks do not depend on the logits; small rooms under the
reference's Bernoulli policy), size-independent invariants elsewhere."""
import numpy as np
import pytest

from conftest import seed_without_near_tie
from learn_region_grow_amd import preprocess, synthetic, workloads
from oracle import grow_ref, rng_ref
from test_gpu_fullsize import check_invariants, zero_net

pytestmark = pytest.mark.gpu
WEIGHT_KW = dict(seed=0, gain=2.0, bias_std=0.2, add_bias_shift=0.0, rmv_bias_shift=-3.0)
CACHE = '/tmp/lrg_cache'


@pytest.fixture(scope='module')
def net(cuda_device):
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    return LrgNetHIP(1, 1, 512, 512, 13, 0, device=cuda_device).load_weights(synthetic.make_synthetic_weights(**WEIGHT_KW))


def gpu_net_fn(net):
    def fn(xi, xn):
        _, add, _, rmv, _ = net.run(xi, xn)
        return add, rmv
    return fn


def small_room(seed, n_raw, furniture=0, room_id=0):
    raw = (synthetic.area5_shaped_room(n_raw, seed, n_furniture=furniture) if furniture
           else synthetic.generate_room_points(n_raw, seed)).astype(np.float32)
    p = preprocess.preprocess_room(raw[:, :6], raw[:, 6].astype(int), raw[:, 7].astype(int))
    return dict(points=p['points'], obj_id=p['obj_id'], order=p['order'], room_id=room_id)


def regions_of(res):
    return [(r['seed'], r['steps'], r['points'], r['reason'], r['labeled']) for r in res.regions]


# ---- configs[3]: 16 restarts per seed, batched ---------------------------------------------------------------------------
def test_sixteen_restarts_batched_match_the_oracle(net):
    """R = 16 restarts of every seed in one launch (G = 16 slots per room), the reference's Bernoulli policy: regions, restart
    step totals and labels equal oracle.grow_room(restarts=16) evaluating the same GPU network (test_random_restart.py:169-197)."""
    from learn_region_grow_amd.grow import RegionGrower
    rooms = [small_room(430 + i, 500 + 250 * i, furniture=2 * i, room_id=60 + i) for i in range(2)]

    def oracle(seed):
        return [grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.CounterStream(seed, room['room_id']),
                                   net_fn=gpu_net_fn(net), restarts=16) for room in rooms]
    seed, wants = seed_without_near_tie(oracle, range(31, 37), 5e-7)
    got = RegionGrower(net, rooms_in_flight=2, rng='counter', seed=seed, restarts=16, group_size=16).run(rooms)
    for res, want in zip(got, wants):
        assert regions_of(res) == regions_of(want)
        np.testing.assert_array_equal(res.cluster_label, want.cluster_label)
        np.testing.assert_array_equal(res.filled_label, want.filled_label)


def test_sixteen_restarts_full_configuration(net):
    """The configuration as benchmarked: the 68-room Area-5-shaped set, 16 restarts per seed batched, 1088 slots in flight.
    Ground-truth masks: every room passes the invariants; two Area-5-size rooms equal the oracle's restart loop exactly."""
    from learn_region_grow_amd.grow import LanedRegionGrower
    rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir=CACHE)
    lg = LanedRegionGrower(net, rooms_in_flight=68, rng='counter', seed=17, policy='gt', restarts=16)
    assert sum(g.S for g in lg.growers) == 68 * 16
    got = lg.run(rooms)
    for room, res in zip(rooms, got):
        check_invariants(room, res)
        assert all(0 <= x['restart'] < 16 for x in res.regions)
    by_size = np.argsort([len(r['points']) for r in rooms])
    for i in (int(by_size[0]), int(by_size[len(by_size) // 2])):        # the smallest and the median-size room (2.2 k / 9.3 k points)
        room = rooms[i]
        want = grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.CounterStream(17, room['room_id']),
                                  net_fn=zero_net, policy='gt', restarts=16)
        assert regions_of(got[i]) == regions_of(want)
        np.testing.assert_array_equal(got[i].filled_label, want.filled_label)


# ---- configs[2]: ScanNet shape ---------------------------------------------------------------------------------------------
def test_scannet_shaped_rooms(net):
    """workloads.scannet_rooms(8), all eight in flight (one per GPU in the named case; here one GPU holds them): ground-truth
    masks exactly equal to the oracle, the Bernoulli policy through the invariants + independence from what else is in flight."""
    from learn_region_grow_amd.grow import LanedRegionGrower, RegionGrower
    rooms = workloads.scannet_rooms(8, seed_base=7000, cache_dir=CACHE)
    got = LanedRegionGrower(net, rooms_in_flight=8, lanes=2, rng='counter', seed=23, policy='gt').run(rooms)
    for room, res in zip(rooms, got):
        want = grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.CounterStream(23, room['room_id']),
                                  net_fn=zero_net, policy='gt')
        assert regions_of(res) == regions_of(want)
        np.testing.assert_array_equal(res.cluster_label, want.cluster_label)
        np.testing.assert_array_equal(res.filled_label, want.filled_label)
        check_invariants(room, res)
    a = RegionGrower(net, rooms_in_flight=8, rng='counter', seed=23, policy='net').run(rooms)
    b = RegionGrower(net, rooms_in_flight=3, rng='counter', seed=23, policy='net').run(rooms[::-1])[::-1]
    for room, x, y in zip(rooms, a, b):
        check_invariants(room, x)
        np.testing.assert_array_equal(x.filled_label, y.filled_label)


# ---- configs[1]: the benchmark configuration --------------------------------------------------------------------------------
@pytest.mark.parametrize('mode', ['free-run', 'lock-step'])
def test_benchmark_configuration_labels(net, mode):
    """68 Area-5-shaped rooms in flight, as bench.py runs them -- free-running launches (one lane), or lock-step iterations over the
    automatic number of lanes with HIP-graph replays: every room passes the invariants; the labels of eight rooms -- the 45 k-point
    one, the smallest, the median and five more -- equal single-room oracle runs."""
    import torch
    from learn_region_grow_amd.grow import LanedRegionGrower
    rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir=CACHE)
    from learn_region_grow_amd.grow import auto_lanes
    if mode == 'free-run':
        lg = LanedRegionGrower(net, rooms_in_flight=68, lanes=None, rng='counter', seed=0, policy='gt')
        assert len(lg.growers) == 1
        got = lg.run(rooms)
        assert lg.growers[0].free_run
    else:
        lg = LanedRegionGrower(net, rooms_in_flight=68, lanes=None, rng='counter', seed=0, policy='gt', graph_iterations=4, free_run=False)
        assert len(lg.growers) == auto_lanes(68) == 3
        got = lg.run(rooms)
        assert all(g.packed and g._graph is not None for g in lg.growers)
    for room, res in zip(rooms, got):
        check_invariants(room, res)
    by_size = [int(i) for i in np.argsort([len(r['points']) for r in rooms])]
    assert len(rooms[by_size[-1]]['points']) > 40000
    picks = [by_size[-1], by_size[0], by_size[len(by_size) // 2]] + by_size[5:60:11]
    for i in picks:
        room = rooms[i]
        want = grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.CounterStream(0, room['room_id']),
                                  net_fn=zero_net, policy='gt')
        assert regions_of(got[i]) == regions_of(want), 'room %d (%d points)' % (i, len(room['points']))
        np.testing.assert_array_equal(got[i].cluster_label, want.cluster_label)
        np.testing.assert_array_equal(got[i].filled_label, want.filled_label)


def test_restarts_scoring_ml_matches_the_oracle(net):
    """test_random_restart.py --scoring ml (as evidently intended): the restart with the largest summed
    log-likelihood of its sampled masks wins.  GPU (logf, fixed-order double sums) against the oracle (NumPy float32 log,
    float64 sum) on the same GPU network; seeds whose winner leads by less than 1e-4 relative are passed over."""
    from learn_region_grow_amd.grow import RegionGrower
    rooms = [small_room(450 + i, 600 + 300 * i, furniture=2, room_id=70 + i) for i in range(2)]

    def oracle(seed):
        out = [grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.CounterStream(seed, room['room_id']),
                                  net_fn=gpu_net_fn(net), restarts=4, scoring='ml') for room in rooms]
        for o in out:                                      # fold the score lead into the margin the seed search looks at
            o.min_rel_margin = min(o.min_rel_margin, 5e-7 * o.min_score_gap / 1e-4)
        return out
    seed, wants = seed_without_near_tie(oracle, range(41, 49), 5e-7)
    got = RegionGrower(net, rooms_in_flight=2, rng='counter', seed=seed, restarts=4, group_size=2, scoring='ml').run(rooms)
    np_scored = RegionGrower(net, rooms_in_flight=2, rng='counter', seed=seed, restarts=4, group_size=2).run(rooms)
    differs = False
    for res, want, other in zip(got, wants, np_scored):
        assert regions_of(res) == regions_of(want)
        np.testing.assert_array_equal(res.filled_label, want.filled_label)
        differs |= not np.array_equal(res.cluster_label, other.cluster_label)
    assert differs, "'ml' and 'np' scoring picked the same restart everywhere: the test rooms do not exercise the score"


@pytest.mark.parametrize('mode,policy', [('lock-step', 'gt'), ('lock-step', 'net'), ('free-run', 'net')])
def test_benchmark_configuration_with_a_busy_chip(net, mode, policy):
    """Slots must not depend on when their workgroups start.  The front kernel allocates the packed rows of an iteration from
    row 0 again, so a slot whose workgroup starts late -- here: while a second stream keeps every CU busy with dense LrgNet
    evaluations -- would find last iteration's rows overwritten if it still looked for them there (it did, up to ABI 3: with two
    lanes 13 of 68 rooms gave more than one outcome over eight runs, tools/determinism_check.py).  The default lanes + the hog, twice,
    against a quiet single-lane run: same regions and labels for all 68 rooms -- under ground-truth masks and under the Bernoulli
    policy that bench.py times, for the lock-step lanes and for the free-running launches (whose workgroups then are not all resident
    from the start: the hog's tiles hold CUs)."""
    import threading
    import torch
    from learn_region_grow_amd.grow import LanedRegionGrower
    rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir=CACHE)
    kw = dict(rooms_in_flight=68, rng='counter', seed=0, policy=policy)
    quiet = LanedRegionGrower(net, lanes=1, free_run=False, **kw).run(rooms)      # (lock-step, one lane: every formulation gives the same bits)
    dev = net.device
    rs = np.random.RandomState(0)
    xi = torch.from_numpy((rs.randn(68, 512, 13) * 0.5).astype(np.float32)).to(dev)
    xn = torch.from_numpy((rs.randn(68, 512, 13) * 0.5).astype(np.float32)).to(dev)
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    hog_net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(synthetic.make_synthetic_weights(**WEIGHT_KW))
    hog_stream = torch.cuda.Stream(device=dev)
    stop = threading.Event()

    def hog():
        with torch.cuda.stream(hog_stream):
            while not stop.is_set():
                for _ in range(8):
                    hog_net.forward(xi, xn)
                hog_stream.synchronize()
    for _ in range(2):
        stop.clear()
        th = threading.Thread(target=hog)
        th.start()
        try:
            if mode == 'lock-step':
                busy = LanedRegionGrower(net, lanes=None, graph_iterations=4, free_run=False, **kw).run(rooms)     # (three lanes)
            else:
                busy = LanedRegionGrower(net, lanes=None, **kw).run(rooms)                                          # (free-running, one lane)
        finally:
            stop.set()
            th.join()
        for i, (a, b) in enumerate(zip(quiet, busy)):
            assert regions_of(a) == regions_of(b), 'room %d (%d points)' % (i, len(rooms[i]['points']))
            np.testing.assert_array_equal(a.cluster_label, b.cluster_label)
            np.testing.assert_array_equal(a.filled_label, b.filled_label)


def test_voxel_grid_and_channel_major_copy_change_nothing(net, monkeypatch):
    """LrgRoom.vgrid (box queries and voxel lookups from a dense grid) and LrgRoom.chan_major (median keys from a channel-major
    copy) are layouts, not definitions: rooms loaded without them give the same regions and labels -- Bernoulli policy, a room
    large enough for regions above 1024 points, the 45 k-point room under ground-truth masks."""
    from learn_region_grow_amd.grow import RegionGrower
    rooms = [small_room(640 + i, n, furniture=f, room_id=80 + i) for i, (n, f) in enumerate([(700, 1), (2600, 3), (6500, 4)])]
    big = workloads.make_room(45063, 1057, 57)

    def run():
        a = RegionGrower(net, rooms_in_flight=3, rng='counter', seed=5, policy='net').run(rooms)
        b = RegionGrower(net, rooms_in_flight=1, rng='counter', seed=5, policy='gt').run([big])
        return a + b
    with_both = run()
    monkeypatch.setenv('LRG_NO_VGRID', '1')
    without_grid = run()
    monkeypatch.setenv('LRG_NO_CHAN_MAJOR', '1')
    without_either = run()
    assert max(r['points'] for res in with_both for r in res.regions) > 1024
    for x, y, z in zip(with_both, without_grid, without_either):
        assert regions_of(x) == regions_of(y) == regions_of(z)
        np.testing.assert_array_equal(x.filled_label, y.filled_label)
        np.testing.assert_array_equal(x.filled_label, z.filled_label)


# ---- configs[1] exactly as bench.py builds it, against the oracle ------------------------------------------------------------
@pytest.mark.parametrize('mode', ['free-run', 'lock-step'])
def test_benchmark_configuration_as_benchmarked_matches_the_oracle(cuda_device, mode):
    """What bench.py times -- the weights trained by this repository (synthetic.load_trained_weights), the reference's Bernoulli
    policy (test_region_grow.py:266-267), 68 rooms in flight, free-running launches (default) or three lock-step lanes with graph
    replays of four iterations -- against single-room oracle runs evaluating the same GPU network: the 45 k-point room, the
    median, the smallest and three more.  The oracle's one-instance evaluation and the loop's give the same logits bit for bit
    (one summation order in every formulation: vector kernel, matrix-core GEMM, free-running pooled blocks); the lock-step case
    evaluates the oracle's network as eight copies all the same, i.e. through the GEMM, to prove that."""
    import torch
    from learn_region_grow_amd.grow import LanedRegionGrower
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    weights = synthetic.load_trained_weights()
    net1 = LrgNetHIP(1, 1, 512, 512, 13, 0, device=cuda_device).load_weights(weights)
    net8 = LrgNetHIP(8, 1, 512, 512, 13, 0, device=cuda_device).load_weights(weights)
    rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir=CACHE)
    by_size = [int(i) for i in np.argsort([len(r['points']) for r in rooms])]
    assert len(rooms[by_size[-1]]['points']) > 40000
    picks = [by_size[-1], by_size[len(by_size) // 2], by_size[0]] + by_size[9:60:20]
    assert len(picks) >= 6

    def net_fn(xi, xn):
        if mode == 'free-run':
            _, add, _, rmv, _ = net1.run(xi, xn)
            return add, rmv
        _, add, _, rmv, _ = net8.run(np.repeat(xi, 8, axis=0), np.repeat(xn, 8, axis=0))
        return add[:1], rmv[:1]

    def oracle(seed):
        return [grow_ref.grow_room(rooms[i]['points'], rooms[i]['obj_id'], rooms[i]['order'], None,
                                   rng_ref.CounterStream(seed, rooms[i]['room_id']), net_fn=net_fn, policy='net', faithful=False) for i in picks]
    seed, wants = seed_without_near_tie(oracle, range(0, 6), 5e-7)
    if mode == 'free-run':
        lg = LanedRegionGrower(net1, rooms_in_flight=68, lanes=None, rng='counter', seed=seed, policy='net')
    else:
        lg = LanedRegionGrower(net1, rooms_in_flight=68, lanes=None, rng='counter', seed=seed, policy='net', graph_iterations=4, free_run=False)
    got = lg.run(rooms)
    assert (len(lg.growers) == 1 and lg.growers[0].free_run) if mode == 'free-run' else (len(lg.growers) == 3 and all(g._graph is not None for g in lg.growers))
    for i, want in zip(picks, wants):
        assert regions_of(got[i]) == regions_of(want), 'room %d (%d points)' % (i, len(rooms[i]['points']))
        np.testing.assert_array_equal(got[i].cluster_label, want.cluster_label)
        np.testing.assert_array_equal(got[i].filled_label, want.filled_label)
    for room, res in zip(rooms, got):                      # (no partition invariant here: a region the network empties leaves its seed to the fill-in)
        assert res.filled_label.min() >= 1 and len(res.regions) > 0
