"""CPU: host-side decisions of the round-5 features (no GPU, no library calls that compute)."""
import numpy as np
import pytest


class _Net:           # what RegionGrower.free_run_applies looks at
    mode = 'fused'
    num_inlier_points = 512
    num_neighbor_points = 512
    lite = 0


def _room(n, extent=5.0, seed=0):
    rs = np.random.RandomState(seed)
    p = np.zeros((n, 13), np.float32)
    p[:, :3] = rs.rand(n, 3) * extent
    return dict(points=p, obj_id=np.zeros(n, np.int32), order=np.arange(n, dtype=np.int32))


def test_auto_speculate_by_rooms_per_gpu():
    from learn_region_grow_amd.grow import auto_speculate
    assert [auto_speculate(n) for n in (1, 8, 16, 17, 32, 33, 68, 272)] == [3, 3, 3, 2, 2, 0, 0, 0]


def test_free_run_applies_matches_the_growers_rules():
    from learn_region_grow_amd import _lib
    from learn_region_grow_amd.grow import RegionGrower
    net = _Net()
    rooms = [_room(2000), _room(3000, seed=1)]
    assert RegionGrower.free_run_applies(net, rooms, 68)
    assert not RegionGrower.free_run_applies(net, rooms, 68, free_run=False)
    assert not RegionGrower.free_run_applies(net, rooms, 68, restarts=16)                      # restarts: lock-step groups
    assert not RegionGrower.free_run_applies(net, rooms, 68, rng='legacy')
    assert not RegionGrower.free_run_applies(net, rooms, 68, packed=False)
    assert not RegionGrower.free_run_applies(net, rooms, _lib.LRG_FREE_RUN_AUTO_SLOTS + 1)     # hundreds of slots: lock-step by default ...
    assert RegionGrower.free_run_applies(net, rooms, _lib.LRG_FREE_RUN_AUTO_SLOTS + 1, free_run=True)      # ... unless asked for
    far = _room(100, extent=250.0)                       # 2 500 voxels across at 0.1 m: no packed voxel words (2048 x 2048 x 1024 at most)
    assert not RegionGrower.free_run_applies(net, [far], 4)
    assert RegionGrower.free_run_applies(net, [far], 4, resolution=0.3)
    lite1 = _Net(); lite1.lite = 1
    assert not RegionGrower.free_run_applies(lite1, rooms, 68)
    assert not RegionGrower.free_run_applies(net, [], 68)


def test_speculate_argument_checks():
    from learn_region_grow_amd.grow import RegionGrower
    with pytest.raises(ValueError):
        RegionGrower(_Net(), speculate=3, restarts=4)
    with pytest.raises(ValueError):
        RegionGrower(_Net(), speculate=3, free_run=False)
    with pytest.raises(ValueError):
        RegionGrower(_Net(), speculate=17)


def test_lanes_refuse_free_running_side_by_side():
    from learn_region_grow_amd.grow import LanedRegionGrower
    with pytest.raises(ValueError):
        LanedRegionGrower(_Net(), rooms_in_flight=8, lanes=2, free_run=True)


def test_tail_and_queue_bytes_are_host_arithmetic(hip_lib):
    assert hip_lib.lrg_grow_async_tail_bytes(0, 4096) == 0 and hip_lib.lrg_grow_async_tail_bytes(68, 100) == 0      # (rows: a multiple of 32)
    assert hip_lib.lrg_grow_async_tail_bytes(68, 4096) == 4 * (32 + 4 * 128 + 2 * 68)
    assert hip_lib.lrg_grow_async_queue_bytes(68) > 0


def test_bench_arguments():
    import bench
    import sys
    old = sys.argv
    try:
        sys.argv = ['bench.py']
        a = bench.parse()
        assert a.speculate == -1 and a.one_room_ks == '1,2,3,4,6' and a.one_rank_collective == 1 and '320' in a.best_slots
    finally:
        sys.argv = old
