"""GPU: several regions of ONE room in flight (LrgAsyncBuffers.speculate, csrc/lrg_front.inl "speculation").

The reference grows a room's regions strictly one after the other (test_region_grow.py:186-188: the next seed is the next UNVISITED point in
curvature order; :227-228: visited points are no candidates; :210-217: a region marks its points when it stops), so a room is one chain of
dependent steps.  With speculate = K the regions of the next K unvisited seeds grow side by side, are committed in seed order, and a region
is dropped and grown again when an earlier commit took a point inside a box it had queried.  The result must be the sequential one: the same
regions in the same order with the same cluster ids, the same filled labels -- compared here with the one-slot-per-room launches and with the
CPU oracle, on rooms where regions do get voided (counted), across launch boundaries (a few steps per launch), and with trained weights."""
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


@pytest.fixture(scope='module')
def trained_net(cuda_device):
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    return LrgNetHIP(1, 1, 512, 512, 13, 0, device=cuda_device).load_weights(synthetic.load_trained_weights())


def _rooms():
    # three ordinary small rooms, two with furniture, and a closet: 350 points in a box so small that every region's dilated box meets
    # every other's -- each commit there voids whatever grows beside it (the forced conflict)
    return [small_room(400 + i, 600 + 200 * i, room_id=10 + i) for i in range(3)] + \
           [small_room(300, 1500, furniture=4, room_id=13), small_room(301, 2500, furniture=6, room_id=14), small_room(77, 350, room_id=15)]


def _same(got, want):
    for g, w in zip(got, want):
        same_regions(g.regions, w.regions)
        np.testing.assert_array_equal(g.cluster_label, w.cluster_label)
        np.testing.assert_array_equal(g.filled_label, w.filled_label)


@pytest.mark.parametrize('K,in_flight,steps,policy', [(2, 6, 64, 'net'), (3, 3, 64, 'net'), (4, 1, 64, 'net'), (8, 2, 64, 'net'), (16, 1, 64, 'net'),
                                                       (3, 2, 1, 'net'), (4, 3, 5, 'net'), (4, 2, 64, 'gt'), (2, 1, 3, 'gt')])
def test_speculation_gives_the_sequential_result(net, K, in_flight, steps, policy):
    """steps: evaluations per slot and launch -- 1, 3, 5: pending and voided regions cross launch boundaries."""
    from learn_region_grow_amd.grow import RegionGrower
    rooms = _rooms()
    kw = dict(rooms_in_flight=in_flight, rng='counter', seed=123, policy=policy)
    want = RegionGrower(net, free_run=True, **kw).run(rooms)
    gr = RegionGrower(net, speculate=K, free_run_steps=steps, **kw)
    got = gr.run(rooms)
    assert gr.free_run and gr.speculate == K and gr.S == in_flight * K
    _same(got, want)
    work = gr.a_work.cpu().numpy()
    committed = sum(r.total_steps for r in got)
    # the closet voids regions for every K; what was thrown away is accounted for: executed = committed + voided evaluations
    assert work[4] > 0 and work[5] > 0
    assert int(work[0]) == committed + int(work[5])
    assert gr.instance_steps == committed + int(work[6])          # the device's step counter = kept steps + the voided regions' steps


def test_speculation_matches_the_oracle(net):
    from learn_region_grow_amd.grow import RegionGrower
    rooms = _rooms()[2:]

    def oracle(seed):
        return [grow_ref.grow_room(room['points'], room['obj_id'], room['order'], None, rng_ref.CounterStream(seed, room['room_id']),
                                   net_fn=gpu_net_fn(net)) for room in rooms]
    seed, wants = seed_without_near_tie(oracle, range(123, 131), SAME_LOGITS_MARGIN)
    for K, in_flight in ((3, 2), (5, 4)):
        gr = RegionGrower(net, rooms_in_flight=in_flight, rng='counter', seed=seed, speculate=K, free_run_steps=32)
        res = gr.run(rooms)
        for i, want in enumerate(wants):
            same_regions(res[i].regions, want.regions)
            np.testing.assert_array_equal(res[i].cluster_label, want.cluster_label)
            np.testing.assert_array_equal(res[i].filled_label, want.filled_label)
        assert int(gr.a_work.cpu().numpy()[4]) > 0        # regions were voided on the way, and the result is still the oracle's


def test_one_room_per_gpu_under_trained_weights(trained_net):
    """The corner BASELINE configs 3 and 5 name: ONE room on the chip.  An Area-5-shaped room under the trained weights and the Bernoulli policy,
    four regions in flight: the sequential result, in fewer launches' worth of chain."""
    from learn_region_grow_amd import workloads
    from learn_region_grow_amd.grow import RegionGrower
    room = dict(workloads.make_room(4000, 1021, 1021), room_id=5)      # (a small Area-5-shaped room: walls, floor, furniture)
    kw = dict(rooms_in_flight=1, rng='counter', seed=11, policy='net')
    want = RegionGrower(trained_net, free_run=True, **kw).run([room])
    for K in (2, 4):
        gr = RegionGrower(trained_net, speculate=K, **kw)
        got = gr.run([room])
        _same(got, want)
        assert len(got[0].regions) > 10
