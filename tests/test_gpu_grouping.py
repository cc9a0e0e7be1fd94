"""GPU: tf_ops/grouping replacements against the C oracle (which equals the reference's CPU functions)."""
import numpy as np
import pytest

from oracle import grouping_ref as G, grow_ref

pytestmark = pytest.mark.gpu


def dev(a, cuda_device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda_device)


@pytest.mark.parametrize('b,n,m,radius,ns', [(2, 200, 50, 0.2, 16), (3, 1024, 256, 0.1, 32), (1, 64, 16, 0.8, 32),
                                              (2, 70, 33, 0.01, 8), (32, 512, 128, 0.1, 64)])
def test_query_ball_point(cuda_device, b, n, m, radius, ns):
    from learn_region_grow_amd import grouping
    rs = np.random.RandomState(n)
    x1 = rs.rand(b, n, 3).astype(np.float32)
    x2 = rs.rand(b, m, 3).astype(np.float32)
    idx, cnt = grouping.query_ball_point(radius, ns, dev(x1, cuda_device), dev(x2, cuda_device))
    widx, wcnt = G.query_ball_point(radius, ns, x1, x2)
    np.testing.assert_array_equal(cnt.cpu().numpy(), wcnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), widx)


def test_group_point_and_grad(cuda_device):
    from learn_region_grow_amd import grouping
    rs = np.random.RandomState(0)
    pts = rs.rand(3, 200, 7).astype(np.float32)
    idx = rs.randint(0, 200, (3, 50, 16)).astype(np.int32)
    out = grouping.group_point(dev(pts, cuda_device), dev(idx, cuda_device))
    np.testing.assert_array_equal(out.cpu().numpy(), G.group_point(pts, idx))
    go = rs.rand(3, 50, 16, 7).astype(np.float32)
    gp = grouping.group_point_grad(dev(pts, cuda_device), dev(idx, cuda_device), dev(go, cuda_device))
    np.testing.assert_allclose(gp.cpu().numpy(), G.group_point_grad(go, idx, 200), rtol=1e-5, atol=1e-5)   # atomic order
    with pytest.raises(ValueError):
        grouping.group_point(dev(pts[0], cuda_device), dev(idx, cuda_device))


def test_selection_sort_known_answer_and_ties(cuda_device):
    from learn_region_grow_amd import grouping
    d = (10 - np.arange(16)).astype(np.float32).reshape(2, 2, 4)      # tf_ops/grouping/test/selection_sort.cpp:65-94
    oi, o = grouping.select_top_k(3, dev(d, cuda_device))
    assert oi.cpu().numpy().reshape(-1).tolist() == [3, 2, 1, 0] * 4
    assert o.cpu().numpy().reshape(-1).tolist() == [7, 8, 9, 10, 3, 4, 5, 6, -1, 0, 1, 2, -5, -4, -3, -2]
    rs = np.random.RandomState(1)
    d = rs.rand(3, 9, 300).astype(np.float32)
    d[0, 0, 5] = d[0, 0, 9] = d[0, 0, 250] = -1.0
    d[1, 2, :] = 0.5
    oi, o = grouping.select_top_k(17, dev(d, cuda_device))
    wi, wo = G.selection_sort(17, d)
    np.testing.assert_array_equal(oi.cpu().numpy(), wi)      # all n positions, not just the first k
    np.testing.assert_array_equal(o.cpu().numpy(), wo)


def test_knn_point(cuda_device):
    from learn_region_grow_amd import grouping
    rs = np.random.RandomState(2)
    x1 = rs.randn(2, 500, 3).astype(np.float32)
    x2 = rs.randn(2, 40, 3).astype(np.float32)
    val, idx = grouping.knn_point(8, dev(x1, cuda_device), dev(x2, cuda_device))
    dist = G.knn_dist(x1, x2)
    wi, wo = G.selection_sort(8, dist)
    np.testing.assert_array_equal(idx.cpu().numpy(), wi[:, :, :8])
    np.testing.assert_array_equal(val.cpu().numpy(), wo[:, :, :8])


@pytest.mark.parametrize('F', [6, 13])
def test_nn1_fill(cuda_device, hip_lib, F):
    import torch
    from learn_region_grow_amd import _lib
    from learn_region_grow_amd.lrgnet import _ptr, _stream_ptr
    rs = np.random.RandomState(F)
    P = (rs.randn(3000, F) * 10 ** rs.uniform(-2, 2, (3000, F))).astype(np.float32)
    P[10] = P[20]
    lab = ((rs.rand(3000) < 0.3) * rs.randint(1, 9, 3000)).astype(np.int32)
    out = torch.zeros(3000, dtype=torch.int32, device=cuda_device)
    dP, dlab = dev(P, cuda_device), dev(lab, cuda_device)      # keep the device buffers alive across the launch
    _lib.check(hip_lib.lrg_nn1_fill(_ptr(dP), 3000, F, _ptr(dlab), _ptr(out), _stream_ptr()), 'nn1')
    np.testing.assert_array_equal(out.cpu().numpy(), grow_ref.fill_unlabeled(P, lab.astype(np.int64)))
