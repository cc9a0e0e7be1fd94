"""CPU: oracle building blocks -- RNG spec, C grouping oracle vs the reference's own C++ (oracle/_ref),
the known-answer vector of tf_ops/grouping/test/selection_sort.cpp, NumPy-order 1-NN fill."""
import ctypes
import os
import sys

import numpy as np
import pytest

from oracle import grouping_ref as G, grow_ref, rng_ref

HAVE_REF = G.ref_lib('query_ball_point') is not None and G.ref_lib('selection_sort') is not None


class quiet_stdout:
    """selection_sort_cpu of the reference prints every element; silence fd 1 around it."""
    def __enter__(self):
        sys.stdout.flush()
        self.fd = os.dup(1)
        self.dn = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.dn, 1)

    def __exit__(self, *a):
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.fd, 1)
        os.close(self.dn)
        os.close(self.fd)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), '6627e8d5 e169c58d bc57ac4c 9b00dbd8'),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, '408f276d 41c83b0e a20bc7c6 6d5451fd'),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            'd16cfe09 94fdcceb 5001e420 24126ea1')]
    for c, k, want in kat:
        r = rng_ref.philox4x32_10(*[np.array([x], dtype=np.uint64) for x in c], k[0], k[1])
        assert ' '.join('%08x' % int(x[0]) for x in r) == want


@pytest.mark.parametrize('n', [2, 3, 17, 512, 513, 4097, 45063])
def test_feistel_is_a_bijection(n):
    p = rng_ref.feistel_permute(np.arange(n), n, [0x12345678, 0x9abcdef0, 0x0fedcba9, 0x87654321])
    assert sorted(p.tolist()) == list(range(n))


def test_counter_stream_sampling_rules():
    s = rng_ref.CounterStream(7, 3)
    ctx = (5, 0, 2)
    p = s.sample(100, 512, rng_ref.PURPOSE_NEIGHBOR, ctx)          # n < k: range(n) then with replacement
    assert p[:100].tolist() == list(range(100)) and p.max() < 100 and len(p) == 512
    q = s.sample(2000, 512, rng_ref.PURPOSE_INLIER, ctx)           # n >= k: without replacement
    assert len(set(q.tolist())) == 512 and q.max() < 2000
    assert np.array_equal(q, s.sample(2000, 512, rng_ref.PURPOSE_INLIER, ctx))       # pure function of the context
    assert not np.array_equal(q, s.sample(2000, 512, rng_ref.PURPOSE_INLIER, (5, 0, 3)))
    u = s.uniform(512, rng_ref.PURPOSE_ADD, ctx)
    assert u.dtype == np.float32 and 0 <= u.min() and u.max() < 1


def test_legacy_stream_is_the_reference_call_sequence():
    a, b = rng_ref.LegacyStream(0), np.random.RandomState(0)
    assert np.array_equal(a.sample(700, 512), b.choice(700, 512, replace=False))
    assert a.sample(9, 512).tolist() == list(range(9)) + list(b.choice(9, 503, replace=True))
    assert np.array_equal(a.uniform(512), b.random_sample(512))


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference at build time)')
def test_grouping_oracle_equals_reference_cpu_functions():
    rs = np.random.RandomState(0)
    x1 = rs.rand(3, 200, 3).astype(np.float32)
    x2 = rs.rand(3, 50, 3).astype(np.float32)
    i0, c0 = G.query_ball_point(0.2, 16, x1, x2)
    i1, _ = G.query_ball_point(0.2, 16, x1, x2, use_reference=True)
    np.testing.assert_array_equal(i0, i1)
    assert c0.min() == 0 and c0.max() <= 16
    pts = rs.rand(3, 200, 7).astype(np.float32)
    np.testing.assert_array_equal(G.group_point(pts, i0), G.group_point(pts, i0, use_reference=True))
    go = rs.rand(3, 50, 16, 7).astype(np.float32)
    np.testing.assert_array_equal(G.group_point_grad(go, i0, 200), G.group_point_grad(go, i0, 200, use_reference=True))
    d = rs.rand(2, 5, 40).astype(np.float32)
    d[0, 0, 5] = d[0, 0, 9] = d[0, 0].min() - 1     # ties: first minimum must win
    with quiet_stdout():
        ri, ro = G.selection_sort(7, d, use_reference=True)
    mi, mo = G.selection_sort(7, d)
    np.testing.assert_array_equal(ri, mi)
    np.testing.assert_array_equal(ro, mo)


def test_selection_sort_known_answer():
    # main() of tf_ops/grouping/test/selection_sort.cpp:65-94: b=2,n=4,m=2,k=3, dist[i]=10-i
    d = (10 - np.arange(16)).astype(np.float32).reshape(2, 2, 4)
    oi, o = G.selection_sort(3, d)
    assert oi.reshape(-1).tolist() == [3, 2, 1, 0] * 4
    assert o.reshape(-1).tolist() == [7, 8, 9, 10, 3, 4, 5, 6, -1, 0, 1, 2, -5, -4, -3, -2]
    if HAVE_REF:
        with quiet_stdout():
            ri, ro = G.selection_sort(3, d, use_reference=True)
        np.testing.assert_array_equal(ri, oi)
        np.testing.assert_array_equal(ro, o)


def test_knn_dist_matches_numpy():
    rs = np.random.RandomState(1)
    x1 = rs.randn(2, 30, 3).astype(np.float32)
    x2 = rs.randn(2, 7, 3).astype(np.float32)
    want = ((x1[:, None, :, :] - x2[:, :, None, :]) ** 2)
    want = (want[..., 0] + want[..., 1]) + want[..., 2]
    np.testing.assert_array_equal(G.knn_dist(x1, x2), want)


@pytest.mark.parametrize('F', [6, 9, 12, 13])
def test_nn1_fill_is_numpy_exact(F):
    rs = np.random.RandomState(F)
    P = (rs.randn(600, F) * 10 ** rs.uniform(-2, 2, (600, F))).astype(np.float32)
    P[10] = P[20]                        # exact duplicate -> tie on distance 0 ... first labeled index wins
    lab = ((rs.rand(600) < 0.3) * rs.randint(1, 9, 600)).astype(np.int64)
    np.testing.assert_array_equal(G.nn1_fill(P, lab), grow_ref.fill_unlabeled(P, lab))
    np.testing.assert_array_equal(G.nn1_fill(P, np.zeros(600, int)), np.zeros(600, int))   # nothing labeled: no-op


def test_voxelize_is_float32_half_even():
    x = np.array([[0.25, 0.35, 0.05], [-0.25, 1.15, 2.5], [0.15, 0.45, -0.05]], dtype=np.float32)
    want = np.round(x / 0.1).astype(int)          # the reference expression (test_region_grow.py:175)
    np.testing.assert_array_equal(grow_ref.voxelize(x, 0.1), want)


def test_host_preprocessing_equals_oracle_loop_on_dense_neighbourhoods():
    """The vectorised host P0 (learn_region_grow_amd.preprocess) against the oracle loop where a point's 27-voxel
    neighbourhood holds ~100-200 raw points: float64 accumulation order matters there (numpy.add.reduceat would not do)."""
    from learn_region_grow_amd import preprocess, synthetic
    from oracle import preprocess_ref
    r = synthetic.area5_shaped_room(1500, 77).astype(np.float32)
    raw = (r[:, :6], r[:, 6].astype(int), r[:, 7].astype(int))
    want = preprocess_ref.preprocess_room(*raw)
    got = preprocess.preprocess_room(*raw, chunk=500)
    assert len(raw[0]) / len(want['points']) > 3            # several raw points per voxel
    for k in ('equalized_idx', 'unequalized_idx', 'obj_id', 'cls_id', 'points', 'curvatures'):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    np.testing.assert_array_equal(got['order'], np.argsort(want['curvatures']))
