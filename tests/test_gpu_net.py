"""GPU: LrgNet forward through the C-ABI (HIP kernels) against the oracle and the reference-made goldens.
Tolerance: |delta| <= 1e-4 + 1e-5*|x| relative to the activation scale of the layer."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from learn_region_grow_amd import synthetic
from oracle import lrgnet_ref

pytestmark = pytest.mark.gpu
WEIGHT_KW = dict(seed=0, gain=2.0, bias_std=0.2, add_bias_shift=0.0, rmv_bias_shift=-3.0)


def close(got, want, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(1.0, float(np.abs(want).max()))
    err = np.abs(got - want)
    tol = 1e-4 + 1e-5 * np.abs(want)         # SURVEY.md 8d, as stated: absolute 1e-4 plus 1e-5 relative, no scale factor
    print('%-16s max |err| %.3e  (max |x| %.3e)' % (what, err.max(), scale))
    assert (err <= tol).all(), '%s: max err %.3e (scale %.3e) at %s' % (what, err.max(), scale, np.unravel_index(err.argmax(), err.shape))


MODES = {'streamed': dict(mode='streamed'), 'streamed+pool': dict(mode='streamed', fuse_pool=True),
         'streamed-tiles': dict(mode='streamed-tiles'), 'fused': dict(mode='fused', keep_acts=True)}


def make_net(cuda_device, lite, F, ni, nn, mode='streamed'):
    import torch
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    w = synthetic.make_synthetic_weights(feature_size=F, lite=lite, **WEIGHT_KW)
    net = LrgNetHIP(1, 1, ni, nn, F, lite, device=cuda_device, **MODES[mode]).load_weights(w)
    return net, w


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'lrgnet_*.npz'))), ids=os.path.basename)
@pytest.mark.parametrize('mode', sorted(MODES))
def test_forward_matches_reference_goldens_layer_by_layer(cuda_device, path, mode):
    import torch
    g = np.load(path)
    lite = int(g['lite'])
    lite = None if lite < 0 else lite
    F = int(g['feature_size'])
    B, ni = g['inlier'].shape[:2]
    nn = g['neighbor'].shape[1]
    if mode == 'fused' and (ni % 64 or nn % 64):
        pytest.skip('the fused kernels tile 64 rows; lrg_forward takes the layer-streamed path for this shape')
    net, w = make_net(cuda_device, lite, F, ni, nn, mode)
    xi = torch.from_numpy(g['inlier']).to(cuda_device)
    xn = torch.from_numpy(g['neighbor']).to(cuda_device)
    add, rmv = net.forward(xi, xn)
    torch.cuda.synchronize()
    nc = len(net.conv_channels)
    for i in range(nc):
        close(net.intermediate('conv', i, B).cpu().numpy().reshape(g['conv%d' % i].shape), g['conv%d' % i], 'conv%d' % i)
        close(net.intermediate('neighbor_conv', i, B).cpu().numpy().reshape(g['neighbor_conv%d' % i].shape),
              g['neighbor_conv%d' % i], 'neighbor_conv%d' % i)
    close(net.intermediate('pooled', 0, B).cpu().numpy().reshape(g['pooled'].shape), g['pooled'], 'pooled')
    for i in range(len(net.conv2_channels)):
        close(net.intermediate('add_hidden', i, B).cpu().numpy().reshape(g['add_conv%d' % i].shape), g['add_conv%d' % i], 'add_conv%d' % i)
        close(net.intermediate('remove_hidden', i, B).cpu().numpy().reshape(g['remove_conv%d' % i].shape), g['remove_conv%d' % i], 'remove_conv%d' % i)
    close(add.cpu().numpy(), g['add_output'], 'add_output')
    close(rmv.cpu().numpy(), g['remove_output'], 'remove_output')
    # the reference-shaped run() wrapper, with the logged scalars
    loss, a, add_acc, r, rmv_acc = net.run(g['inlier'], g['neighbor'], g['add_mask'], g['rmv_mask'])
    np.testing.assert_allclose(loss, g['loss'], rtol=1e-4)
    assert abs(add_acc - g['add_acc']) <= 1.0 / nn + 1e-6 and abs(rmv_acc - g['remove_acc']) <= 1.0 / ni + 1e-6


@pytest.mark.parametrize('B', [1, 3, 68])
@pytest.mark.parametrize('mode', sorted(MODES))
def test_forward_full_size_against_oracle(cuda_device, B, mode):
    """The shape the loop uses: 512 inliers + 512 neighbours x 13 features (test_region_grow.py:22-24)."""
    import torch
    net, w = make_net(cuda_device, 0, 13, 512, 512, mode)
    rs = np.random.RandomState(B)
    xi = (rs.randn(B, 512, 13) * 0.5).astype(np.float32)
    xn = (rs.randn(B, 512, 13) * 0.5).astype(np.float32)
    xi[:, 300:] = xi[:, :212]          # duplicated rows, as the padding rule produces (:240)
    add, rmv = net.forward(torch.from_numpy(xi).to(cuda_device), torch.from_numpy(xn).to(cuda_device))
    take = slice(0, min(B, 3))
    wadd, wrmv = lrgnet_ref.forward(w, xi[take], xn[take], dtype=np.float64)
    # instances are independent: the first few must equal the oracle run on them alone
    close(add.cpu().numpy()[take], wadd, 'add')
    close(rmv.cpu().numpy()[take], wrmv, 'rmv')
    if B > 3:
        wadd, wrmv = lrgnet_ref.forward(w, xi[-1:], xn[-1:], dtype=np.float64)
        close(add.cpu().numpy()[-1:], wadd, 'add[-1]')
        close(rmv.cpu().numpy()[-1:], wrmv, 'rmv[-1]')
    # duplicate rows give identical logits; permuting rows permutes logits (max-pool is order-free)
    r = rmv.cpu().numpy()
    np.testing.assert_array_equal(r[:, 300:], r[:, :212])
    perm = rs.permutation(512)
    add2, rmv2 = net.forward(torch.from_numpy(xi[:, perm].copy()).to(cuda_device), torch.from_numpy(xn).to(cuda_device))
    np.testing.assert_allclose(rmv2.cpu().numpy(), r[:, perm], rtol=0, atol=1e-4 * max(1, np.abs(r).max()))


def test_single_layer_entry_points(cuda_device, hip_lib):
    import ctypes
    import torch
    from learn_region_grow_amd.lrgnet import _ptr, _stream_ptr
    from learn_region_grow_amd import _lib
    rs = np.random.RandomState(3)
    for rows, cin, cout in [(128, 13, 64), (200, 64, 128), (64, 128, 512), (77, 64, 2), (130, 12, 64)]:
        x = rs.randn(rows, cin).astype(np.float32)
        w = rs.randn(cin, cout).astype(np.float32)
        b = rs.randn(cout).astype(np.float32)
        dx, dw, db = [torch.from_numpy(a).to(cuda_device) for a in (x, w, b)]
        dy = torch.empty((rows, cout), dtype=torch.float32, device=cuda_device)
        _lib.check(hip_lib.lrg_pointwise_layer(_ptr(dx), cin, _ptr(dw), cout, _ptr(db), _ptr(dy), rows, cin, cout, 1, 0, 0,
                                               None, 0, _stream_ptr()), 'lrg_pointwise_layer')
        want = np.maximum(x.astype(np.float64) @ w.astype(np.float64) + b, 0)
        close(dy.cpu().numpy(), want, 'layer %dx%dx%d' % (rows, cin, cout))
    # segmax
    x = rs.randn(5, 40, 96).astype(np.float32)
    dx = torch.from_numpy(x).to(cuda_device)
    out = torch.zeros((5, 200), dtype=torch.float32, device=cuda_device)
    _lib.check(hip_lib.lrg_segmax(_ptr(dx), ctypes.c_void_p(out.data_ptr() + 4 * 100), 5, 40, 96, 200, _stream_ptr()), 'lrg_segmax')
    np.testing.assert_array_equal(out.cpu().numpy()[:, 100:196], x.max(axis=1))
    # bad arguments are rejected, not launched
    assert hip_lib.lrg_pointwise_layer(None, 4, None, 4, None, None, 4, 4, 4, 1, 0, 0, None, 0, None) <= -1000


def test_fused_and_streamed_paths_agree(cuda_device):
    """Same weights, same inputs: the 3-launch fused evaluation and the layer-streamed one give the same logits to
    fp32 rounding (the k-order inside a dot product is the only difference)."""
    import torch
    rs = np.random.RandomState(5)
    xi = torch.from_numpy((rs.randn(9, 512, 13) * 0.5).astype(np.float32)).to(cuda_device)
    xn = torch.from_numpy((rs.randn(9, 512, 13) * 0.5).astype(np.float32)).to(cuda_device)
    outs = {}
    for mode in ('streamed', 'fused', 'streamed-tiles'):
        net, _ = make_net(cuda_device, 0, 13, 512, 512, mode)
        add, rmv = net.forward(xi, xn)
        outs[mode] = (add.cpu().numpy(), rmv.cpu().numpy(), net.intermediate('pooled', 0, 9).cpu().numpy().copy())
    close(outs['fused'][0], outs['streamed'][0], 'add')
    close(outs['fused'][1], outs['streamed'][1], 'rmv')
    close(outs['fused'][2], outs['streamed'][2], 'pooled')
    # one layer per launch on the fused tile: the fused stacks' own MFMA sequence and epilogue, so the pooled features (a max of identical products) are the same bits
    close(outs['streamed-tiles'][0], outs['streamed'][0], 'add (tiles)')
    close(outs['streamed-tiles'][1], outs['streamed'][1], 'rmv (tiles)')
    np.testing.assert_array_equal(outs['streamed-tiles'][2], outs['fused'][2])


@pytest.mark.parametrize('B,ni,nn', [(1, 64, 64), (3, 64, 192), (5, 256, 128), (2, 512, 512), (37, 128, 64)])
def test_streaming_layer_kernel_on_ragged_problem_sizes(cuda_device, B, ni, nn):
    """The layer-streamed evaluation on the streaming wavefront kernel (csrc/lrg_stream_layer.inl: both branches / both heads in one launch per layer, each a
    problem of its own row count, per-instance bias rows in the heads' first layer) against the round-1 layer kernel: the same logits to fp32 rounding, every
    intermediate too -- few tiles per column group, row counts that differ between the two problems, instances of one or two tiles."""
    import torch
    rs = np.random.RandomState(100 * B + ni + nn)
    xi = torch.from_numpy((rs.randn(B, ni, 13) * 0.5).astype(np.float32)).to(cuda_device)
    xn = torch.from_numpy((rs.randn(B, nn, 13) * 0.5).astype(np.float32)).to(cuda_device)
    outs = {}
    for mode in ('streamed', 'streamed-tiles'):
        net, _ = make_net(cuda_device, 0, 13, ni, nn, mode)
        add, rmv = net.forward(xi, xn)
        torch.cuda.synchronize()
        outs[mode] = dict(add=add.cpu().numpy().copy(), rmv=rmv.cpu().numpy().copy(), pooled=net.intermediate('pooled', 0, B).cpu().numpy().copy())
        for i in range(len(net.conv_channels)):
            outs[mode]['conv%d' % i] = net.intermediate('conv', i, B).cpu().numpy().copy()
            outs[mode]['nconv%d' % i] = net.intermediate('neighbor_conv', i, B).cpu().numpy().copy()
    assert outs['streamed']['add'].shape == (B, nn, 2) and outs['streamed']['rmv'].shape == (B, ni, 2)
    for k in sorted(outs['streamed']):
        close(outs['streamed-tiles'][k], outs['streamed'][k], k)


def test_forward_rows_skips_duplicate_rows_exactly(cuda_device):
    """Sets padded by duplication (test_region_grow.py:240,:252): evaluating only the distinct leading rows gives
    bit-identical logits for them and the same pooled feature as evaluating all 512 rows."""
    import torch
    rs = np.random.RandomState(8)
    B = 7
    n_in = np.array([1, 57, 64, 65, 300, 512, 0])
    n_nb = np.array([5, 512, 128, 63, 449, 200, 17])
    xi = (rs.randn(B, 512, 13) * 0.5).astype(np.float32)
    xn = (rs.randn(B, 512, 13) * 0.5).astype(np.float32)
    for b in range(B):
        for x, n in ((xi, n_in[b]), (xn, n_nb[b])):
            if 0 < n < 512:
                x[b, n:] = x[b, rs.randint(0, n, 512 - n)]
    net, _ = make_net(cuda_device, 0, 13, 512, 512, 'fused')
    dxi, dxn = torch.from_numpy(xi).to(cuda_device), torch.from_numpy(xn).to(cuda_device)
    add_full, rmv_full = [t.cpu().numpy().copy() for t in net.forward(dxi, dxn)]
    pooled_full = net.intermediate('pooled', 0, B).cpu().numpy().copy().reshape(B, -1)
    add = torch.full((B, 512, 2), float('nan'), device=cuda_device)
    rmv = torch.full((B, 512, 2), float('nan'), device=cuda_device)
    rin = torch.from_numpy(n_in.astype(np.int32)).to(cuda_device)
    rnb = torch.from_numpy(n_nb.astype(np.int32)).to(cuda_device)
    net.forward(dxi, dxn, add, rmv, rows_in=rin, rows_nb=rnb)
    add, rmv = add.cpu().numpy(), rmv.cpu().numpy()
    pooled = net.intermediate('pooled', 0, B).cpu().numpy().reshape(B, -1)
    for b in range(B):
        if n_in[b] == 0 or n_nb[b] == 0:
            continue                      # a zero count skips the instance: outputs undefined
        np.testing.assert_array_equal(rmv[b, :n_in[b]], rmv_full[b, :n_in[b]])
        np.testing.assert_array_equal(add[b, :n_nb[b]], add_full[b, :n_nb[b]])
        np.testing.assert_array_equal(pooled[b], pooled_full[b])
        tiles = -(-n_in[b] // 32) * 32
        assert np.isnan(rmv[b, tiles:]).all()          # whole tiles of duplicates were never touched


@pytest.mark.gpu
@pytest.mark.parametrize('lite,F', [(0, 13), (1, 13), (2, 12), (0, 6)])
def test_packed_weight_image(cuda_device, hip_lib, lite, F):
    """lrg_pack_weights lays every kernel out in MFMA-operand order: image[((cb*ng+g)*64+lane)*4+s] =
    W[8g+4(lane>>5)+s][32cb+(lane&31)] with Cin zero-padded to a multiple of 8; and a forward with w->packed = NULL
    (image rebuilt in the workspace on every call) gives the same bits as one with the cached image."""
    import torch
    net, weights = make_net(cuda_device, lite, F, 128, 128, 'fused')
    img = net._packed.view(torch.float32).cpu().numpy()
    cc, c2 = net.conv_channels, net.conv2_channels
    off = 0
    layers = []
    for pre in ('lrg_', 'lrg_neighbor_'):
        for i in range(len(cc)):
            layers.append(np.asarray(weights['%skernel%d' % (pre, i)], np.float32)[0])
    for pre in ('lrg_add_', 'lrg_remove_'):
        for j in range(len(c2)):
            W = np.asarray(weights['%skernel%d' % (pre, j)], np.float32)[0]
            layers.append(W[2 * cc[-1]:] if j == 0 else W)
    for W in layers:
        K, N = W.shape
        ng, ncb = (K + 7) // 8, (N + 31) // 32
        Wp = np.zeros((ng * 8, ncb * 32), np.float32)
        Wp[:K, :N] = W
        # [cb][g][lh][li][s]  <-  Wp[8g + 4lh + s][32cb + li]
        want = Wp.reshape(ng, 2, 4, ncb, 32).transpose(3, 0, 1, 4, 2).reshape(-1)
        np.testing.assert_array_equal(img[off:off + want.size], want)
        off = -(-(off + want.size) // 64) * 64
    assert off * 4 == net._packed.numel()
    rs = np.random.RandomState(11)
    xi = torch.from_numpy((rs.randn(5, 128, F) * 0.5).astype(np.float32)).to(cuda_device)
    xn = torch.from_numpy((rs.randn(5, 128, F) * 0.5).astype(np.float32)).to(cuda_device)
    a0, r0 = [t.cpu().numpy().copy() for t in net.forward(xi, xn)]
    keep = net._w.packed
    net._w.packed = None
    a1, r1 = [t.cpu().numpy().copy() for t in net.forward(xi, xn)]
    net._w.packed = keep
    np.testing.assert_array_equal(a0, a1)
    np.testing.assert_array_equal(r0, r1)


@pytest.mark.parametrize('lite,F', [(0, 13), (1, 13), (2, 12)])
def test_forward_packed_equals_dense_rows(cuda_device, lite, F):
    """lrg_forward_packed (the loop's formulation: the distinct rows of all instances back to back, dense 32-row tiles that
    span instances, max-pool and hoisted bias per run of rows) gives the bits of lrg_forward_rows on the same rows: ragged
    row counts including 0 (skipped instance), 1, tile multiples and full sets."""
    import torch
    net, w = make_net(cuda_device, lite, F, 512, 512, 'fused')
    rs = np.random.RandomState(5)
    rows_in = np.array([57, 1, 0, 512, 33, 64, 7, 300, 31, 1, 2, 96], dtype=np.int32)
    rows_nb = np.array([26, 200, 0, 512, 5, 1, 64, 17, 480, 3, 1, 32], dtype=np.int32)
    B = len(rows_in)
    xi = (rs.randn(B, 512, F) * 0.5).astype(np.float32)
    xn = (rs.randn(B, 512, F) * 0.5).astype(np.float32)
    for b in range(B):                      # the padding rule (:240,:252): rows past the count are copies of earlier ones
        if rows_in[b]:
            xi[b, rows_in[b]:] = xi[b, rs.randint(0, rows_in[b], 512 - rows_in[b])]
        if rows_nb[b]:
            xn[b, rows_nb[b]:] = xn[b, rs.randint(0, rows_nb[b], 512 - rows_nb[b])]
    dxi, dxn = torch.from_numpy(xi).to(cuda_device), torch.from_numpy(xn).to(cuda_device)
    add_d, rmv_d = net.forward(dxi, dxn, rows_in=torch.from_numpy(rows_in).to(cuda_device), rows_nb=torch.from_numpy(rows_nb).to(cuda_device))
    add_d, rmv_d = add_d.cpu().numpy(), rmv_d.cpu().numpy()
    cap = B * 512
    order = rs.permutation(B)               # instances in arbitrary order in the packed arrays (allocation order is not slot order)
    pin, pnb = np.zeros((cap, F), np.float32), np.zeros((cap, F), np.float32)
    rin, rnb = np.full(cap, -7, np.int32), np.full(cap, -7, np.int32)
    off_in, off_nb, oi, on = {}, {}, 0, 0
    for b in order:
        off_in[b], off_nb[b] = oi, on
        pin[oi:oi + rows_in[b]] = xi[b, :rows_in[b]]; rin[oi:oi + rows_in[b]] = b; oi += rows_in[b]
        pnb[on:on + rows_nb[b]] = xn[b, :rows_nb[b]]; rnb[on:on + rows_nb[b]] = b; on += rows_nb[b]
    pin[oi:] = np.nan                       # whatever lies past the counts must not matter
    pnb[on:] = np.nan
    t = lambda a: torch.from_numpy(a).to(cuda_device)
    add_p, rmv_p, pooled = net.forward_packed(t(pin), t(pnb), t(rin), t(rnb), t(np.array([oi, on], np.int32)), B)
    torch.cuda.synchronize()
    add_p, rmv_p, pooled = add_p.cpu().numpy(), rmv_p.cpu().numpy(), pooled.cpu().numpy()
    for b in range(B):
        np.testing.assert_array_equal(add_p[off_nb[b]:off_nb[b] + rows_nb[b]], add_d[b, :rows_nb[b]], err_msg='add, instance %d' % b)
        np.testing.assert_array_equal(rmv_p[off_in[b]:off_in[b] + rows_in[b]], rmv_d[b, :rows_in[b]], err_msg='rmv, instance %d' % b)
    # the pooled features against the oracle's max over the distinct rows
    C = net.conv_channels[-1]
    for b in range(B):
        if rows_in[b] == 0:
            assert not pooled[b].any()
            continue
        _, _, acts = lrgnet_ref.forward(w, xi[b:b + 1, :rows_in[b]], xn[b:b + 1, :rows_nb[b]], lite=lite, dtype=np.float64,
                                        return_acts=True)
        close(pooled[b], acts['pooled'][0], 'pooled, instance %d' % b)
