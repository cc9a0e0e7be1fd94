"""GPU: tf_ops/grouping replacements against the C oracle (which equals the reference's CPU functions)."""
import numpy as np
import pytest

from oracle import grouping_ref as G, grow_ref

pytestmark = pytest.mark.gpu


def dev(a, cuda_device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda_device)


@pytest.mark.parametrize('b,n,m,radius,ns', [(2, 200, 50, 0.2, 16), (3, 1024, 256, 0.1, 32), (1, 64, 16, 0.8, 32),
                                              (2, 70, 33, 0.01, 8), (32, 512, 128, 0.1, 64),
                                              (2, 1000, 20, 0.5, 100), (1, 1000, 20, 0.5, 300), (1, 300, 9, 0.45, 64)])      # (lists that fill up inside / behind a round of four chunks)
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


@pytest.mark.parametrize('b,n,m,c,k', [(2, 500, 40, 3, 8), (3, 2048, 33, 3, 128), (1, 64, 5, 3, 64), (2, 1300, 17, 6, 40), (1, 4096, 9, 3, 70),
                                        (2, 70, 64, 1, 1), (1, 2048, 5, 3, 300), (1, 1024, 4, 3, 512)])
def test_knn_point_fused_and_unfused(cuda_device, b, n, m, c, k):
    """knn_point (tf_grouping.py:48-73): the fused lrg_knn_topk (distances + selection in registers) and the reference's three steps
    (distance matrix, select_top_k, slice) against the C oracle's selection sort -- same swap sequence, so the same order among
    EQUAL distances too (duplicate dataset points, a grid of coordinates)."""
    from learn_region_grow_amd import grouping
    rs = np.random.RandomState(n + k)
    x1 = rs.randn(b, n, c).astype(np.float32)
    x2 = rs.randn(b, m, c).astype(np.float32)
    x1[:, 5] = x1[:, 9]                                    # exact ties: duplicated dataset points ...
    x1[:, n // 2] = x1[:, 3]
    if n >= 500:
        x1[0, :64] = np.round(x1[0, :64])                  # ... and a lattice: many equal distances to a lattice query
        x2[0, 0] = 0.0
    dist = G.knn_dist(x1, x2)
    wi, wo = G.selection_sort(k, dist)
    for fused in (True, False):
        val, idx = grouping.knn_point(k, dev(x1, cuda_device), dev(x2, cuda_device), fused=fused)
        np.testing.assert_array_equal(idx.cpu().numpy(), wi[:, :, :k], err_msg='fused=%s' % fused)
        np.testing.assert_array_equal(val.cpu().numpy(), wo[:, :, :k], err_msg='fused=%s' % fused)


@pytest.mark.parametrize('b,n,m,ns,c', [(5, 300, 40, 16, 16), (9, 512, 33, 32, 24), (2, 512, 128, 64, 64), (3, 100, 20, 16, 8), (1, 2000, 64, 8, 40)])
def test_group_point_grad_workgroup_owned_slices(cuda_device, b, n, m, ns, c):
    """The gradient's LDS formulation (c a multiple of 8, m x nsample >= 256: a workgroup owns 8 channels of a batch item, the workgroups of an item on one XCD;
    batch sizes that are no multiple of 8) against the C oracle, with index lists padded by copies of their first entry as the ball query pads them
    (tf_grouping_g.cu:20-23) and with the op's accumulate-into semantics (the caller's zeroed buffer)."""
    from learn_region_grow_amd import grouping
    rs = np.random.RandomState(b * 1000 + c)
    pts = rs.rand(b, n, c).astype(np.float32)
    idx = rs.randint(0, n, (b, m, ns)).astype(np.int32)
    keep = rs.randint(1, ns + 1, (b, m))
    for bi in range(b):
        for j in range(m):
            idx[bi, j, keep[bi, j]:] = idx[bi, j, 0]
    go = rs.randn(b, m, ns, c).astype(np.float32)
    gp = grouping.group_point_grad(dev(pts, cuda_device), dev(idx, cuda_device), dev(go, cuda_device))
    want = G.group_point_grad(go, idx, n)
    np.testing.assert_allclose(gp.cpu().numpy(), want, rtol=2e-5, atol=2e-5 * ns)      # (sums of up to m x nsample terms in another order)


def test_more_neighbours_than_the_register_selection_holds(cuda_device):
    """k > 512: the selection with the row in registers is instantiated for the first eight register rows -- select_top_k takes the memory-resident
    kernel, knn_point the reference's three steps, lrg_knn_topk itself says so."""
    from learn_region_grow_amd import grouping, _lib
    rs = np.random.RandomState(11)
    x1 = rs.randn(1, 1024, 3).astype(np.float32)
    x2 = rs.randn(1, 3, 3).astype(np.float32)
    dist = G.knn_dist(x1, x2)
    wi, wo = G.selection_sort(600, dist)
    oi, o = grouping.select_top_k(600, dev(dist, cuda_device))
    np.testing.assert_array_equal(oi.cpu().numpy(), wi)
    np.testing.assert_array_equal(o.cpu().numpy(), wo)
    val, idx = grouping.knn_point(600, dev(x1, cuda_device), dev(x2, cuda_device))
    np.testing.assert_array_equal(idx.cpu().numpy(), wi[:, :, :600])
    np.testing.assert_array_equal(val.cpu().numpy(), wo[:, :, :600])
    with pytest.raises(_lib.LrgHipError):
        grouping.knn_point(600, dev(x1, cuda_device), dev(x2, cuda_device), fused=True)


def test_selection_sort_full_rows_large(cuda_device):
    """select_top_k keeps the reference op's full-size outputs: every one of the n positions (values and indices) after k passes
    equals the oracle's, at the harness shape of tf_ops/grouping/test/selection_sort.cu (n = 2048, k = 128), with NaN / inf entries."""
    from learn_region_grow_amd import grouping
    rs = np.random.RandomState(7)
    d = rs.rand(2, 24, 2048).astype(np.float32)
    d[0, 1, 3] = np.nan
    d[0, 1, 700] = np.inf
    d[0, 2, :] = np.inf
    d[0, 3, 0] = np.nan
    d[1, 0, 100:140] = 0.25
    oi, o = grouping.select_top_k(128, dev(d, cuda_device))
    wi, wo = G.selection_sort(128, d)
    np.testing.assert_array_equal(oi.cpu().numpy(), wi)
    np.testing.assert_array_equal(o.cpu().numpy(), wo)


@pytest.mark.parametrize('b,n,m,ns,c', [(4, 512, 128, 64, 64), (3, 100, 7, 5, 12), (1, 33, 1, 1, 4), (2, 257, 13, 9, 20)])
def test_group_point_wide_rows(cuda_device, b, n, m, ns, c):
    """c a multiple of 4: the 16-byte path (the harness shape of tf_ops/grouping/test/query_ball_point.cpp: c = 64; sizes that end inside a
    workgroup's two pieces per thread)."""
    from learn_region_grow_amd import grouping
    rs = np.random.RandomState(3)
    pts = rs.rand(b, n, c).astype(np.float32)
    idx = rs.randint(0, n, (b, m, ns)).astype(np.int32)
    out = grouping.group_point(dev(pts, cuda_device), dev(idx, cuda_device))
    np.testing.assert_array_equal(out.cpu().numpy(), G.group_point(pts, idx))


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


@pytest.mark.parametrize('F,n,frac', [(13, 3000, 0.3), (6, 5000, 0.9), (12, 1037, 0.02), (9, 64, 0.5), (13, 2500, 0.0), (13, 700, 1.0)])
def test_nn1_fill_tiled(cuda_device, hip_lib, F, n, frac):
    """lrg_nn1_fill_ws (tiled, 64-bit atomicMin on (distance bits, index)) equals the oracle and the one-workgroup-per-point
    kernel: same float32 distances, first minimum on ties; nothing labeled -> all zero; everything labeled -> copy."""
    import torch
    from learn_region_grow_amd import _lib
    from learn_region_grow_amd.lrgnet import _ptr, _stream_ptr
    rs = np.random.RandomState(n + F)
    P = (rs.randn(n, F) * 10 ** rs.uniform(-2, 2, (n, F))).astype(np.float32)
    P[n // 3] = P[n // 2]                                   # an exact tie between two candidates
    P[5] = P[6]
    lab = ((rs.rand(n) < frac) * rs.randint(1, 9, n)).astype(np.int32)
    want = grow_ref.fill_unlabeled(P, lab.astype(np.int64)) if 0 < (lab != 0).sum() else lab.astype(np.int64)
    dP, dlab = dev(P, cuda_device), dev(lab, cuda_device)
    out = torch.full((n,), -7, dtype=torch.int32, device=cuda_device)
    ws = torch.empty(hip_lib.lrg_nn1_fill_workspace_bytes(n), dtype=torch.uint8, device=cuda_device)
    for _ in range(2):                                      # the workspace is reusable
        _lib.check(hip_lib.lrg_nn1_fill_ws(_ptr(dP), n, F, _ptr(dlab), _ptr(out), _ptr(ws), ws.numel(), _stream_ptr()), 'nn1 ws')
        np.testing.assert_array_equal(out.cpu().numpy(), want)
    ref = torch.zeros(n, dtype=torch.int32, device=cuda_device)
    _lib.check(hip_lib.lrg_nn1_fill(_ptr(dP), n, F, _ptr(dlab), _ptr(ref), _stream_ptr()), 'nn1')
    np.testing.assert_array_equal(ref.cpu().numpy(), want)
    assert hip_lib.lrg_nn1_fill_ws(_ptr(dP), n, F, _ptr(dlab), _ptr(out), _ptr(ws), 16, _stream_ptr()) <= -1000


def test_nn1_fill_tiled_on_a_room(cuda_device, hip_lib):
    """Points in object order, as rooms come (the pruned pairs of lrg_nn1_fill_ws: most chunks are farther from a query in xyz than its
    best so far): the result is that of the exhaustive one-workgroup-per-point search, ties included (duplicated rows), with whole
    objects unlabeled and a stretch of points without any labeled neighbour nearby."""
    import torch
    from learn_region_grow_amd import _lib, workloads
    from learn_region_grow_amd.lrgnet import _ptr, _stream_ptr
    room = workloads.area5_rooms(3, seed_base=1000, cache_dir='/tmp/lrg_cache')[2]
    P = np.ascontiguousarray(room['points'][:40000], dtype=np.float32)
    n, F = P.shape
    rs = np.random.RandomState(11)
    obj = np.asarray(room['obj_id'])[:n]
    lab = (obj % 7 + 1).astype(np.int32)
    lab[rs.rand(n) < 0.15] = 0                               # scattered unlabeled points
    for o in np.unique(obj)[::5]:
        lab[obj == o] = 0                                    # whole objects unlabeled
    lab[1000:3000] = 0
    P[5000:5064] = P[17000:17064]                            # exact ties between far-apart indices
    P[123] = P[124]
    dP, dlab = dev(P, cuda_device), dev(lab, cuda_device)
    out = torch.full((n,), -7, dtype=torch.int32, device=cuda_device)
    ref = torch.zeros(n, dtype=torch.int32, device=cuda_device)
    ws = torch.empty(hip_lib.lrg_nn1_fill_workspace_bytes(n), dtype=torch.uint8, device=cuda_device)
    _lib.check(hip_lib.lrg_nn1_fill_ws(_ptr(dP), n, F, _ptr(dlab), _ptr(out), _ptr(ws), ws.numel(), _stream_ptr()), 'nn1 ws')
    _lib.check(hip_lib.lrg_nn1_fill(_ptr(dP), n, F, _ptr(dlab), _ptr(ref), _stream_ptr()), 'nn1')
    np.testing.assert_array_equal(out.cpu().numpy(), ref.cpu().numpy())
    assert (out.cpu().numpy() != 0).all()


def test_nn1_fill_batch_equals_room_by_room(cuda_device, hip_lib):
    """lrg_nn1_fill_batch over 70 rooms of different sizes (two groups of launches; an empty room, a room without unlabeled points, a room
    without labeled points among them) = lrg_nn1_fill_ws room by room."""
    import ctypes
    import torch
    from learn_region_grow_amd import _lib
    from learn_region_grow_amd.lrgnet import _ptr, _stream_ptr
    rs = np.random.RandomState(5)
    F = 13
    base = [3000, 64, 1, 0, 777, 5000, 257, 256, 255, 1200, 90, 4100, 33, 2048, 600, 17, 999, 1500, 320]
    sizes = base + [s + 5 * j if s else 0 for j in (1, 2, 3) for s in base[:17]]      # 70 rooms: two groups of launches (64 rooms per group)
    P, L, O, R = [], [], [], []
    for k, n in enumerate(sizes):
        pts = (rs.randn(max(n, 1), F) * 10 ** rs.uniform(-1, 1, (max(n, 1), F))).astype(np.float32)[:n]
        lab = ((rs.rand(n) < (0.0 if k % 19 == 4 else 1.0 if k % 19 == 6 else 0.4)) * rs.randint(1, 9, n)).astype(np.int32)
        P.append(dev(pts.reshape(n, F) if n else np.zeros((1, F), np.float32), cuda_device)); L.append(dev(lab if n else np.zeros(1, np.int32), cuda_device))
        O.append(torch.full((max(n, 1),), -7, dtype=torch.int32, device=cuda_device)); R.append(torch.full((max(n, 1),), -7, dtype=torch.int32, device=cuda_device))
    jobs = (_lib.LrgFillJob * len(sizes))()
    for k, n in enumerate(sizes):
        jobs[k].points, jobs[k].label_in, jobs[k].label_out, jobs[k].n = P[k].data_ptr(), L[k].data_ptr(), O[k].data_ptr(), n
    need = hip_lib.lrg_nn1_fill_batch_workspace_bytes(jobs, len(sizes))
    ws = torch.empty(need, dtype=torch.uint8, device=cuda_device)
    _lib.check(hip_lib.lrg_nn1_fill_batch(jobs, len(sizes), F, _ptr(ws), ws.numel(), _stream_ptr()), 'batch')
    ws1 = torch.empty(hip_lib.lrg_nn1_fill_workspace_bytes(max(sizes)), dtype=torch.uint8, device=cuda_device)
    for k, n in enumerate(sizes):
        if n:
            _lib.check(hip_lib.lrg_nn1_fill_ws(_ptr(P[k]), n, F, _ptr(L[k]), _ptr(R[k]), _ptr(ws1), ws1.numel(), _stream_ptr()), 'ws')
            np.testing.assert_array_equal(O[k].cpu().numpy()[:n], R[k].cpu().numpy()[:n])
    assert hip_lib.lrg_nn1_fill_batch(jobs, len(sizes), F, _ptr(ws), 64, _stream_ptr()) <= -1000


@pytest.mark.parametrize('b,n,m,radius,ns,c', [(2, 200, 50, 0.2, 16, 7), (3, 1024, 256, 0.1, 32, 64), (1, 64, 16, 0.8, 32, 4), (2, 70, 33, 0.01, 8, 0),
                                                (32, 512, 128, 0.1, 64, 64), (1, 1000, 20, 0.5, 300, 12), (1, 300, 9, 0.45, 64, 3)])
def test_query_ball_group_equals_the_three_ops(cuda_device, b, n, m, radius, ns, c):
    """lrg_query_ball_group = query_ball_point + group_point(xyz) + group_point(points) of sample_and_group (train_pointnet.py:113-121) in one launch: the index
    lists and counts of the C oracle (= the reference's C++), the gathered rows bit for bit, the translation normalization (:117) as a float32 subtraction."""
    from learn_region_grow_amd import grouping
    rs = np.random.RandomState(n + c)
    x1 = rs.rand(b, n, 3).astype(np.float32)
    x2 = rs.rand(b, m, 3).astype(np.float32)
    pts = rs.rand(b, n, c).astype(np.float32) if c else None
    d1, d2 = dev(x1, cuda_device), dev(x2, cuda_device)
    dp = dev(pts, cuda_device) if c else None
    widx, wcnt = G.query_ball_point(radius, ns, x1, x2)
    for sub in (False, True):
        idx, cnt, gx, gp = grouping.query_ball_group(radius, ns, d1, d2, dp, subtract_center=sub)
        np.testing.assert_array_equal(cnt.cpu().numpy(), wcnt)
        np.testing.assert_array_equal(idx.cpu().numpy(), widx)
        want_x = G.group_point(x1, widx)
        if sub:
            want_x = want_x - x2[:, :, None, :]
        np.testing.assert_array_equal(gx.cpu().numpy(), want_x)
        if c:
            np.testing.assert_array_equal(gp.cpu().numpy(), G.group_point(pts, widx))
            np.testing.assert_array_equal(gp.cpu().numpy(), grouping.group_point(dp, idx).cpu().numpy())
        else:
            assert gp is None
    with pytest.raises(ValueError):
        grouping.query_ball_group(radius, ns, d1, d2, dev(np.zeros((b, n + 1, 4), np.float32), cuda_device))
