"""GPU: the training step (learn_region_grow_amd.train.LrgNetTrainer: forward through lrg_forward, backward through lrg_gemm_f32 /
lrg_ce_grad / lrg_pool_backward / lrg_segment_colsum, lrg_adam_step) against the float64 oracle (oracle/train_ref.py), which is
itself pinned to the reference's loss graph and to central differences (tests/test_train_oracle.py)."""
import numpy as np
import pytest

from learn_region_grow_amd import stage, synthetic, workloads
from oracle import train_ref

pytestmark = pytest.mark.gpu
WEIGHT_KW = dict(seed=0, gain=2.0, bias_std=0.2, add_bias_shift=0.0, rmv_bias_shift=-3.0)


def batch(rs, B, N, F):
    xi, xn = (rs.randn(B, N, F) * 0.5).astype(np.float32), (rs.randn(B, N, F) * 0.5).astype(np.float32)
    xi[0, N // 2:] = xi[0, rs.randint(0, N // 2, N - N // 2)]          # a padded set: duplicated rows tie in the max-pool
    am, rm = rs.randint(0, 2, (B, N)).astype(np.int32), (rs.rand(B, N) < 0.2).astype(np.int32)
    return xi, xn, am, rm


def test_gemm_transposes_epilogue_and_split(cuda_device, hip_lib):
    """lrg_gemm_f32: the four operand layouts, odd sizes, addend + ReLU-mask epilogue, split reduction."""
    import ctypes
    import torch
    from learn_region_grow_amd.lrgnet import _ptr, _stream_ptr
    rs = np.random.RandomState(0)
    for (M, N, K, tA, tB, split) in [(70, 50, 37, 0, 0, 1), (64, 128, 200, 1, 0, 1), (33, 2, 900, 1, 0, 4), (300, 64, 2, 0, 1, 1),
                                     (13, 64, 5000, 1, 0, 5), (129, 65, 64, 0, 1, 1)]:
        A = rs.randn(*((K, M) if tA else (M, K))).astype(np.float32)
        Bm = rs.randn(*((N, K) if tB else (K, N))).astype(np.float32)
        add = rs.randn(M, N).astype(np.float32)
        mask = rs.randn(M, N).astype(np.float32)
        want = (A.T if tA else A).astype(np.float64) @ (Bm.T if tB else Bm).astype(np.float64)
        dA, dB = torch.from_numpy(A).to(cuda_device), torch.from_numpy(Bm).to(cuda_device)
        C = torch.zeros((M, N), dtype=torch.float32, device=cuda_device)
        use_epi = split == 1
        dadd, dmask = torch.from_numpy(add).to(cuda_device), torch.from_numpy(mask).to(cuda_device)
        rc = hip_lib.lrg_gemm_f32(M, N, K, _ptr(dA), A.shape[1], tA, _ptr(dB), Bm.shape[1], tB, _ptr(C), N, _ptr(dadd) if use_epi else None,
                                  _ptr(dmask) if use_epi else None, split, _stream_ptr())
        assert rc == 0
        if use_epi:
            want = np.where(mask > 0, want + add, 0.0)
        np.testing.assert_allclose(C.cpu().numpy(), want, rtol=2e-5, atol=2e-4 * np.sqrt(K))
    assert hip_lib.lrg_gemm_f32(4, 4, 4, _ptr(dA), 4, 0, _ptr(dB), 4, 0, _ptr(C), 4, _ptr(dadd), None, 2, _stream_ptr()) <= -1000


@pytest.mark.parametrize('lite,F', [(0, 13), (2, 12), (1, 13)])
def test_gradients_match_the_oracle(cuda_device, lite, F):
    from learn_region_grow_amd.train import LrgNetTrainer
    B, N = 3, 64
    w = synthetic.make_synthetic_weights(feature_size=F, lite=lite, **WEIGHT_KW)
    tr = LrgNetTrainer(B, N, N, F, lite, device=cuda_device).load_weights(w)
    xi, xn, am, rm = batch(np.random.RandomState(lite + 1), B, N, F)
    sc = tr.backward(xi, xn, am, rm)
    loss, G, want = train_ref.loss_and_grads(w, xi, xn, am, rm, lite=lite)
    np.testing.assert_allclose(sc['loss'], loss, rtol=2e-5)
    assert abs(sc['add_acc'] - want['add_acc']) < 1e-9 + 1.0 / (B * N) and abs(sc['remove_acc'] - want['remove_acc']) < 1e-9 + 1.0 / (B * N)
    got = tr.grads_numpy()
    for k in sorted(G):
        scale = max(1e-6, float(np.abs(G[k]).max()))
        err = float(np.abs(got[k] - G[k]).max())
        print('%-24s max |g| %.3e  max err %.3e' % (k, scale, err))
        assert err <= 2e-4 * scale + 1e-6, k


def test_adam_steps_follow_the_oracle(cuda_device):
    """Three train_step calls = three oracle steps (float64 gradients, TensorFlow-1 Adam): the variables stay together."""
    from learn_region_grow_amd.train import LrgNetTrainer
    B, N, F = 2, 64, 13
    w = synthetic.make_synthetic_weights(**WEIGHT_KW)
    tr = LrgNetTrainer(B, N, N, F, 0, device=cuda_device).load_weights(w)
    opt = train_ref.Adam(1e-3)
    rs = np.random.RandomState(9)
    wo = {k: np.asarray(v, np.float32) for k, v in w.items()}
    for step in range(3):
        xi, xn, am, rm = batch(rs, B, N, F)
        loss_o, G, _ = train_ref.loss_and_grads(wo, xi, xn, am, rm)
        loss_g = tr.train_step(xi, xn, am, rm)[0]
        np.testing.assert_allclose(loss_g, loss_o, rtol=1e-4)
        wo = opt.step(wo, G)
    wg = tr.weights_numpy()
    for k in wo:
        # Adam normalises the step: an entry whose gradient is float32 noise around zero can move by up to ~3 lr per step in either
        # direction, so single entries may part by a few lr; almost all stay within float32 rounding of the oracle
        assert float(np.abs(wg[k] - wo[k]).max()) <= 1e-2, k
        assert float(np.mean(np.abs(wg[k] - wo[k]) > 2e-5)) < 0.02, k


def test_training_on_staged_tuples_learns(cuda_device):
    """The loop of train_region_grow.py:141-183 on tuples staged from two synthetic rooms: the loss falls well below its start."""
    from learn_region_grow_amd.train import LrgNetTrainer
    parts = [stage.stage_room(r['points'], r['obj_id'], np.random.RandomState(i)) for i, r in
             enumerate([workloads.make_room(2500, 4000 + i, i) for i in range(2)])]
    data = stage.center_tuples(stage.merge(parts))
    B = 20
    tr = LrgNetTrainer(B, 512, 512, 13, 0, device=cuda_device).load_weights(synthetic.make_reference_init_weights(seed=0))
    rs = np.random.RandomState(0)
    losses = []
    for epoch in range(3):
        idx = np.arange(len(data['points']))
        rs.shuffle(idx)                                                                    # train_region_grow.py:141-142
        for b in range(len(idx) // B):
            xi, xn, ia, ir = train_ref.assemble_batch(data['points'], data['remove'], data['neighbor_points'], data['add'], idx[b * B:(b + 1) * B], rs,
                                                      batch_size=B)
            losses.append(tr.train_step(xi, xn, ia, ir)[0])
    first, last = np.mean(losses[:5]), np.mean(losses[-5:])
    print('loss %.3f -> %.3f over %d steps' % (first, last, len(losses)))
    assert np.isfinite(losses).all() and last < 0.6 * first
