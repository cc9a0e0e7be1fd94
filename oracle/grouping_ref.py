"""ctypes doors onto the C oracle (oracle/liblrg_oracle.so) and, where built, onto the
reference's own CPU functions (oracle/_ref/*.so).  TEST INFRASTRUCTURE, see oracle/__init__.py."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int)


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


def _load(path):
    if not os.path.exists(path):
        return None
    return ctypes.CDLL(path)


def lib():
    p = os.path.join(HERE, 'liblrg_oracle.so')
    if not os.path.exists(p):
        build()
    return ctypes.CDLL(p)


def ref_lib(which):
    return _load(os.path.join(HERE, '_ref', 'libref_%s.so' % which))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def query_ball_point(radius, nsample, xyz1, xyz2, use_reference=False):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.zeros((b, m, nsample), dtype=np.int32)
    cnt = np.zeros((b, m), dtype=np.int32)
    if use_reference:
        ref_lib('query_ball_point').refcpu_query_ball_point(
            b, n, m, ctypes.c_float(radius), nsample, xyz1.ctypes.data_as(_F), xyz2.ctypes.data_as(_F), idx.ctypes.data_as(_I))
        return idx, None
    lib().ref_query_ball_point(b, n, m, ctypes.c_float(radius), nsample, xyz1.ctypes.data_as(_F),
                               xyz2.ctypes.data_as(_F), idx.ctypes.data_as(_I), cnt.ctypes.data_as(_I))
    return idx, cnt


def group_point(points, idx, use_reference=False):
    points, idx = _f(points), _i(idx)
    b, n, c = points.shape
    m, ns = idx.shape[1:]
    out = np.zeros((b, m, ns, c), dtype=np.float32)
    fn = ref_lib('query_ball_point').refcpu_group_point if use_reference else lib().ref_group_point
    fn(b, n, c, m, ns, points.ctypes.data_as(_F), idx.ctypes.data_as(_I), out.ctypes.data_as(_F))
    return out


def group_point_grad(grad_out, idx, n, use_reference=False):
    grad_out, idx = _f(grad_out), _i(idx)
    b, m, ns, c = grad_out.shape
    gp = np.zeros((b, n, c), dtype=np.float32)
    fn = ref_lib('query_ball_point').refcpu_group_point_grad if use_reference else lib().ref_group_point_grad
    fn(b, n, c, m, ns, grad_out.ctypes.data_as(_F), idx.ctypes.data_as(_I), gp.ctypes.data_as(_F))
    return gp


def selection_sort(k, dist, use_reference=False):
    dist = _f(dist)
    b, m, n = dist.shape
    outi = np.zeros((b, m, n), dtype=np.int32)
    out = np.zeros((b, m, n), dtype=np.float32)
    fn = ref_lib('selection_sort').refcpu_selection_sort if use_reference else lib().ref_selection_sort
    fn(b, n, m, k, dist.ctypes.data_as(_F), outi.ctypes.data_as(_I), out.ctypes.data_as(_F))
    return outi, out


def knn_dist(xyz1, xyz2):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    dist = np.zeros((b, m, n), dtype=np.float32)
    lib().ref_knn_dist(b, n, m, c, xyz1.ctypes.data_as(_F), xyz2.ctypes.data_as(_F), dist.ctypes.data_as(_F))
    return dist


def nn1_fill(points, label):
    points = _f(points)
    label = _i(label)
    out = np.zeros_like(label)
    lib().ref_nn1_fill(points.shape[0], points.shape[1], points.ctypes.data_as(_F), label.ctypes.data_as(_I),
                       out.ctypes.data_as(_I))
    return out
