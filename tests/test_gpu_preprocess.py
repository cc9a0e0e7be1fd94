"""GPU: room preprocessing P0 (lrg_preprocess) against the oracle loop (oracle/preprocess_ref.py, a restatement of
test_region_grow.py:119-173).

eig='lapack': integer outputs, covariances and therefore every feature are bit-identical to the oracle.
eig='jacobi': the 3x3 decomposition differs from LAPACK's by ~1e-16 |cov|.  Stated tolerance: unit normals within 1e-6
where the two smallest singular values are separated by more than 1e-5 of the largest (where they are not, the direction is
not defined by the data); normalised curvature within 1e-9 absolute (float64), so the float32 feature column agrees to 1 ulp;
seed order equal wherever neighbouring curvatures differ by more than 1e-9."""
import numpy as np
import pytest

from learn_region_grow_amd import preprocess, synthetic
from oracle import preprocess_ref

pytestmark = pytest.mark.gpu


def raw_room(seed, n=2500, wlh=(1.6, 1.3, 1.0)):
    r = synthetic.generate_room_points(n, seed, wlh=wlh).astype(np.float32)
    return r[:, :6], r[:, 6].astype(int), r[:, 7].astype(int)


@pytest.mark.parametrize('seed,F', [(1, 13), (2, 12), (3, 9), (4, 6)])
def test_lapack_mode_is_bit_exact(cuda_device, seed, F):
    from learn_region_grow_amd import preprocess_gpu
    raw, obj, cls = raw_room(seed)
    want = preprocess_ref.preprocess_room(raw, obj, cls, feature_size=F)
    got = preprocess_gpu.preprocess_room(raw, obj, cls, feature_size=F, eig='lapack', device=cuda_device)
    np.testing.assert_array_equal(got['equalized_idx'], want['equalized_idx'])
    np.testing.assert_array_equal(got['unequalized_idx'], want['unequalized_idx'])
    np.testing.assert_array_equal(got['obj_id'], want['obj_id'])
    np.testing.assert_array_equal(got['cls_id'], want['cls_id'])
    assert got['points'].dtype == np.float32 and got['points'].shape == want['points'].shape
    np.testing.assert_array_equal(got['points'], want['points'])
    np.testing.assert_array_equal(got['curvatures'], want['curvatures'])
    np.testing.assert_array_equal(got['order'], np.argsort(want['curvatures']))


def _check_jacobi(got, want, raw, resolution=0.1):
    np.testing.assert_array_equal(got['equalized_idx'], want['equalized_idx'])
    np.testing.assert_array_equal(got['unequalized_idx'], want['unequalized_idx'])
    np.testing.assert_array_equal(got['obj_id'], want['obj_id'])
    np.testing.assert_array_equal(got['points'][:, :9], want['points'][:, :9])            # xyz, room coordinates, rgb: exact
    assert np.abs(got['curvatures'] - want['curvatures']).max() < 1e-9
    ulp = np.spacing(np.abs(want['points'][:, 12]).astype(np.float32))
    assert (np.abs(got['points'][:, 12] - want['points'][:, 12]) <= ulp).all()
    # normals: compare where the direction is determined
    cov = preprocess.preprocess_room(raw[0], raw[1], raw[2], resolution=resolution, return_cov=True)['cov']
    S = np.linalg.svd(cov, compute_uv=False)
    ok = (S[:, 1] - S[:, 2]) > 1e-5 * S[:, 0]
    assert ok.mean() > 0.9
    assert np.abs(got['points'][ok, 9:12] - want['points'][ok, 9:12]).max() < 1e-6
    # seed order: identical except inside runs of (nearly) equal curvature
    cs = np.sort(want['curvatures'])
    go, wo = got['order'], np.argsort(want['curvatures'])
    diff = np.nonzero(go != wo)[0]
    for k in diff:
        near = cs[min(k + 1, len(cs) - 1)] - cs[max(k - 1, 0)]
        assert near < 1e-9, (k, near)
    return len(diff)


@pytest.mark.parametrize('seed', [1, 5])
def test_jacobi_mode_within_stated_tolerance(cuda_device, seed):
    from learn_region_grow_amd import preprocess_gpu
    raw = raw_room(seed)
    want = preprocess_ref.preprocess_room(*raw)
    got = preprocess_gpu.preprocess_room(*raw, eig='jacobi', device=cuda_device)
    _check_jacobi(got, want, raw)


def test_full_size_room_against_oracle(cuda_device):
    """An Area-5-sized room (86 k raw points -> 20 k equalised, ~150 raw points per neighbourhood) against the oracle loop:
    lapack mode bit-exact, jacobi mode within the tolerance above."""
    from learn_region_grow_amd import preprocess_gpu
    r = synthetic.area5_shaped_room(20000, 1234).astype(np.float32)
    raw = (r[:, :6], r[:, 6].astype(int), r[:, 7].astype(int))
    want = preprocess_ref.preprocess_room(*raw)
    want['order'] = np.argsort(want['curvatures'])
    got = preprocess_gpu.preprocess_room(*raw, eig='lapack', device=cuda_device)
    for k in ('equalized_idx', 'unequalized_idx', 'obj_id', 'cls_id', 'points', 'curvatures', 'order'):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    got = preprocess_gpu.preprocess_room(*raw, eig='jacobi', device=cuda_device)
    _check_jacobi(got, want, raw)


def test_degenerate_inputs(cuda_device, hip_lib):
    from learn_region_grow_amd import _lib, preprocess_gpu
    # many raw points in one voxel, isolated voxels, a voxel-straddling pair: equalisation and order of the lists
    rs = np.random.RandomState(0)
    raw = np.zeros((400, 6), np.float32)
    raw[:300, :3] = 0.5 + rs.rand(300, 3) * 0.04            # one crowded voxel
    raw[300:, :3] = rs.rand(100, 3) * 3                     # scattered
    raw[:, 3:6] = rs.rand(400, 3)
    obj = np.arange(400) % 7
    want = preprocess_ref.preprocess_room(raw, obj, obj)
    got = preprocess_gpu.preprocess_room(raw, obj, obj, eig='lapack', device=cuda_device)
    np.testing.assert_array_equal(got['equalized_idx'], want['equalized_idx'])
    np.testing.assert_array_equal(got['unequalized_idx'], want['unequalized_idx'])
    np.testing.assert_array_equal(got['points'], want['points'])          # includes NaN curvature rows of isolated points
    # a point outside the +-2^20 voxel window is reported, not mis-hashed
    far = raw.copy()
    far[5, 0] = 3e5
    with pytest.raises(_lib.LrgHipError):
        preprocess_gpu.preprocess_room(far, obj, obj, device=cuda_device)
    assert hip_lib.lrg_preprocess_workspace_bytes(0) == 0


@pytest.mark.parametrize('seed,F', [(1, 13), (2, 12), (5, 13), (7, 13)])
def test_exact_mode_equals_the_reference_where_the_loop_reads(cuda_device, seed, F):
    """eig='exact': the Jacobi solve on the GPU with the points whose float32 features or seed-order position could differ under LAPACK
    redone with LAPACK -- features and seed order (all the region-grow loop reads, test_region_grow.py:172,183) equal the oracle bit for
    bit, and only a small share of the points needed the host."""
    from learn_region_grow_amd import preprocess_gpu
    raw, obj, cls = raw_room(seed)
    want = preprocess_ref.preprocess_room(raw, obj, cls, feature_size=F)
    got = preprocess_gpu.preprocess_room(raw, obj, cls, feature_size=F, eig='exact', device=cuda_device)
    for k in ('equalized_idx', 'unequalized_idx', 'obj_id', 'cls_id', 'points'):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    np.testing.assert_array_equal(got['order'], np.argsort(want['curvatures']))
    assert np.abs(got['curvatures'] - want['curvatures']).max() < 1e-12
    assert got['exact_stats']['lapack_points'] <= 0.15 * got['exact_stats']['points'] + 8, got['exact_stats']      # (small rooms: runs of equal curvature)


def test_exact_mode_full_size_and_degenerate_rooms(cuda_device):
    """The 20 k-point Area-5-shaped room, three more Area-5-shaped rooms, and the degenerate input (isolated points: NaN curvatures, a
    crowded voxel) -- points and order bit for bit."""
    from learn_region_grow_amd import preprocess_gpu
    rooms = [synthetic.area5_shaped_room(20000, 1234), synthetic.area5_shaped_room(9000, 77), synthetic.area5_shaped_room(5000, 78),
             synthetic.area5_shaped_room(3000, 79)]
    for r in rooms:
        r = r.astype(np.float32)
        raw = (r[:, :6], r[:, 6].astype(int), r[:, 7].astype(int))
        want = preprocess.preprocess_room(*raw)                      # (the host version: itself bit-equal to the oracle loop, tests/test_oracle_golden.py)
        got = preprocess_gpu.preprocess_room(*raw, eig='exact', device=cuda_device)
        np.testing.assert_array_equal(got['points'], want['points'])
        np.testing.assert_array_equal(got['order'], want['order'])
        np.testing.assert_array_equal(got['obj_id'], want['obj_id'])
        assert got['exact_stats']['lapack_points'] <= 0.15 * got['exact_stats']['points'] + 8, got['exact_stats']      # (small rooms: runs of equal curvature)
    rs = np.random.RandomState(0)
    raw = np.zeros((400, 6), np.float32)
    raw[:300, :3] = 0.5 + rs.rand(300, 3) * 0.04
    raw[300:, :3] = rs.rand(100, 3) * 3
    raw[:, 3:6] = rs.rand(400, 3)
    obj = np.arange(400) % 7
    want = preprocess_ref.preprocess_room(raw, obj, obj)
    got = preprocess_gpu.preprocess_room(raw, obj, obj, eig='exact', device=cuda_device)
    np.testing.assert_array_equal(got['points'], want['points'])          # (NaN rows included: assert_array_equal treats NaN == NaN)
