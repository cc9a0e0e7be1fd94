// A 32-row tile of one LrgNet branch (learn_region_grow_util.py:106-123: five 1x1 convolutions 13 -> 64 -> 64 -> 64 -> 128 -> 512, bias, ReLU, max-pool) by single
// wavefronts, activations in registers.  Round 6; included by lrg_async.inl.
//
// lrg_fused_tile.inl runs such a tile on a team of four wavefronts that hand every layer's output to each other through LDS: a team barrier, an LDS round trip and an
// epilogue of ~600 instructions per pass with ONE wave per SIMD -- 52 k cycles for 23 k cycles of MFMA issue under load (profiles/r05_bench_debug_68.log), and the
// branch stage is the longest piece of a slot's step.  Here the product is computed TRANSPOSED, H_l^T = W_l^T H_(l-1)^T:
//
//   v_mfma_f32_32x32x2_f32   A = the layer's kernel (row i = output channel, from LDS in lrg_pack_weights' operand order: the lane's float4 of k-group g holds
//                                k = 8g + 4h + 0..3 of column 32 cb + lane % 32 -- the image the team tiles read as their B operand, unchanged),
//                            B = the activations (column j = point = lane % 32),
//                            D: lane (point j, half h) holds channels 32 cb + 8 (r >> 2) + 4 h + (r & 3), r = 0 .. 15.
//
// So register 4 (g & 3) + s of output block g >> 2 of a lane holds, in lane half h, channel 8g + 4h + s: exactly the B operand of instruction s of k-group g of the
// NEXT layer ("lane half h feeds logical k = 8g + 4h + s", lrg_fused_tile.inl: tile_mfma).  Bias and ReLU are applied to the accumulators in place and the next
// layer's MFMAs read them: no LDS round trip, no barrier, nothing shared between wavefronts.  Same sums in the same order as the team tile -- per output a chain of
// FMAs over k = 8g + 0, 4, 1, 5, 2, 6, 3, 7 (the instruction adds lane half 0, then lane half 1: tools/mfma_order_check.py), products commute -- so conv[1], the pooled
// maxima and everything downstream are the team tile's bit for bit (tools/wave_tile_probe.hip against a host chain: tests/test_gpu_wave_tile.py; regions and labels
// against the lock-step iterations: tests/test_gpu_free_run.py).
//
// What a wavefront cannot hold is the pooled layer's kernel (128 x 512 floats), and what a CU's LDS cannot hold is all of a branch (331 KB).  Two stages:
//   PREFIX task (tile): layers 0 - 3 (272 MFMAs) on a CU that keeps the four kernels of BOTH branches in LDS (2 x 69 KB); conv[1] (for the heads, :130,:134) and the
//     128-channel output go to HBM rows (write-through, 16-byte stores), then the wavefront publishes the tile's four POOL tasks.
//   POOL task (tile, quarter of the pooled layer's 512 columns): the tile's 128 channels back into registers (sixteen 16-byte loads per lane = the B operands), 2 x 2
//     column blocks of 128 MFMAs each from a CU that keeps its (side, half) of the pooled layer's kernel in LDS (128 KB), the max over the tile's points, one
//     atomicMax instruction per pair of blocks.  The last POOL task of a slot's evaluation publishes its pooled product and head tiles (lrg_async_branch_arrive).
// A k-group's kernel operand is one ds_read_b128 per 4 MFMAs (256 cycles); the matrix pipe of the wave's SIMD is what a task waits for.
// (First form of the round, profiles/r06_wave_probe.txt: ONE stage, a task = (tile, quarter) that ran layers 0 - 3 again -- 528 MFMAs, 46 k cycles = 19.5 us a task,
//  62 % more matrix work per tile; bit-identical too, but with 68 rooms in flight the wave-branch CUs, the head teams and the units were all above 80 % busy and the
//  step got no shorter: 850 k against 879 k instance-steps/s.)
//
// The max-pool (:122-123) of a block of 32 channels: max over the 32 lanes of a half of max(acc + bias, 0) = max(max_lanes(acc) + bias, 0) (rounding is monotone), by a
// TRANSPOSING reduction on the DPP network -- at every level half of the registers are exchanged for the partner lane's other half, 16 registers x 32 lanes -> one
// register in which lane t of a row of 16 holds channel register t: 46 instructions instead of 80, and the 64 maxima of two blocks end up in 64 different lanes: ONE
// atomicMax instruction (on the non-negative float bits, as in the team tile) per pair of blocks.
#pragma once
#include "lrg_fused_tile.inl"

// LDS of a PREFIX CU, in floats from the start of the launch's dynamic LDS: side s at s * LRG_WA_SIDE
#define LRG_WA_W0 0                    // layer 0: 2 column blocks x 2 k-groups x 64 lanes x 4
#define LRG_WA_W1 1024                 // layer 1: 2 x 8 x 256
#define LRG_WA_W2 5120                 // layer 2: 2 x 8 x 256
#define LRG_WA_W3 9216                 // layer 3: 4 x 8 x 256
#define LRG_WA_B0 17408                // biases: 64, 64, 64, 128
#define LRG_WA_B1 17472
#define LRG_WA_B2 17536
#define LRG_WA_B3 17600
#define LRG_WA_SIDE 17728
#define LRG_WA_FLOATS (2 * LRG_WA_SIDE)
// LDS of a POOL CU of (side, half): quarter q of the pooled layer (q = 2 half + 0 / 1) at (q & 1) * LRG_WP_QUARTER
#define LRG_WP_W4 0                    // 4 column blocks x 16 k-groups x 256
#define LRG_WP_B4 16384                // 128
#define LRG_WP_QUARTER 16512
#define LRG_WP_FLOATS (2 * LRG_WP_QUARTER)
#define LRG_WB_FLOATS (LRG_WA_FLOATS > LRG_WP_FLOATS ? LRG_WA_FLOATS : LRG_WP_FLOATS)      // (the fill-in team of such a CU lives behind it)

#ifndef LRG_WB_SMEM
#define LRG_WB_SMEM lrg_async_smem
#endif

// the shape this file is written for (checked on the host: lrg_wave_branch_fits)
#define LRG_WB_K0 16
#define LRG_WB_C0 64
#define LRG_WB_C1 64
#define LRG_WB_C2 64
#define LRG_WB_C3 128
#define LRG_WB_C4 512

__device__ __forceinline__ float4 lrg_wb_lds4(int off_floats) { return *reinterpret_cast<const float4 *>(&LRG_WB_SMEM[off_floats]); }

// acc0 / acc1 (two 32-channel output blocks of a layer) over NG k-groups; h = the previous layer's output blocks (NG / 4 of them); w0 / w1 = float offset of the
// lane's float4 of k-group 0 of the two blocks' kernels.  The operands of group g + 1 are requested before the MFMAs of group g.
template <int NG, int NBI>
__device__ __forceinline__ void lrg_wb_mfma2(f32x16 &acc0, f32x16 &acc1, const f32x16 (&h)[NBI], int w0, int w1) {
    static_assert(NBI * 4 == NG, "four k-groups per input block");
    float4 a0 = lrg_wb_lds4(w0), a1 = lrg_wb_lds4(w1);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float4 n0 = a0, n1 = a1;
        if (g + 1 < NG) { n0 = lrg_wb_lds4(w0 + (g + 1) * 256); n1 = lrg_wb_lds4(w1 + (g + 1) * 256); }
        const f32x16 &hb = h[g >> 2];
        const int r = 4 * (g & 3);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, hb[r + 0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, hb[r + 0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, hb[r + 1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, hb[r + 1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, hb[r + 2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, hb[r + 2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, hb[r + 3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, hb[r + 3], acc1, 0, 0, 0);
        a0 = n0; a1 = n1;
        __builtin_amdgcn_sched_barrier(0);      // (nothing moves across a k-group: left to itself the scheduler requests a whole pass's operands up front -- 128 VGPRs)
    }
}

// bias + ReLU on an output block in place: the lane's channels are 8 j + 4 h + (0 .. 3), j = 0 .. 3 -- four float4 of the bias row
__device__ __forceinline__ void lrg_wb_bias_relu(f32x16 &a, int bias_off /* floats: bias of the block's channel 0 + 4 h */) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 b = lrg_wb_lds4(bias_off + 8 * j);
        a[4 * j + 0] = fmaxf(a[4 * j + 0] + b.x, 0.f);
        a[4 * j + 1] = fmaxf(a[4 * j + 1] + b.y, 0.f);
        a[4 * j + 2] = fmaxf(a[4 * j + 2] + b.z, 0.f);
        a[4 * j + 3] = fmaxf(a[4 * j + 3] + b.w, 0.f);
    }
}

// ---- the transposing max over the 32 lanes of each wave half ----
// one level: the lanes whose bit `sel` is clear keep x and take the partner's x, the others keep y and take the partner's y (partner = DPP control CTRL, which maps
// a lane with the bit clear to one with it set and back)
template <int CTRL>
__device__ __forceinline__ float lrg_wb_level(float x, float y, bool hi) {
    const float keep = hi ? y : x, send = hi ? x : y;
    const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xF, 0xF, false));
    return fmaxf(keep, recv);
}
// 16 registers x 16 lanes of a row -> one register: lane t of the row holds the maximum over the row's lanes of register t
__device__ __forceinline__ float lrg_wb_rowmax16(const f32x16 &a, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    float l1[8];          // registers r and r + 4 (r = 0 .. 3, 8 .. 11): partner lane t ^ 7 (row_half_mirror), selected by lane bit 2
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (i & 3) + 8 * (i >> 2);
        l1[i] = lrg_wb_level<0x141>(a[r], a[r + 4], b2);
    }
    float l2[4];          // r and r + 1: partner t ^ 1 (quad_perm 1, 0, 3, 2), lane bit 0.  l1[i]: r = (i & 3) + 8 (i >> 2)
#pragma unroll
    for (int i = 0; i < 4; ++i) l2[i] = lrg_wb_level<0xB1>(l1[2 * i], l1[2 * i + 1], b0);
    float l3[2];          // r and r + 2: partner t ^ 2 (quad_perm 2, 3, 0, 1), lane bit 1.  l2[i]: r = 2 (i & 1) + 8 (i >> 1)
#pragma unroll
    for (int i = 0; i < 2; ++i) l3[i] = lrg_wb_level<0x4E>(l2[2 * i], l2[2 * i + 1], b1);
    // r and r + 8: partner t ^ 8 (row_ror 8), lane bit 3
    return lrg_wb_level<0x128>(l3[0], l3[1], b3);
}

// PREFIX task: rows r0 .. r0 + 31 of `x` (64-byte stride, uncentred) of slot `slot`, layers 0 - 3 with the kernels at LDS offset `wa` (the side's).
//   x, center: as LrgFusedProb (write-through by the front workgroup: read past the L1); conv1: layer 1's HBM copy [rows, 64]; h3: layer 3's [rows, 128]
__device__ __forceinline__ void lrg_wave_prefix_tile(const float *x, const float *center, float *conv1, float *h3out, long r0, int slot, int wa, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    // ---- the rows: lane (point li, half lh) holds k = 8g + 4 lh + 0..3 of its point for g = 0, 1 (columns past the feature count are zeros on both sides) ----
    f32x16 h0[1];
    {
        const float4 x0 = lrg_ld_coh4(x + r0 * 16, (unsigned)(li * 16 + 4 * lh) * 4u), x1 = lrg_ld_coh4(x + r0 * 16, (unsigned)(li * 16 + 8 + 4 * lh) * 4u);
        const float4 c0 = lrg_ld_coh4(center + (long)slot * 16, (unsigned)(4 * lh) * 4u), c1 = lrg_ld_coh4(center + (long)slot * 16, (unsigned)(8 + 4 * lh) * 4u);
        h0[0][0] = __fsub_rn(x0.x, c0.x); h0[0][1] = __fsub_rn(x0.y, c0.y); h0[0][2] = __fsub_rn(x0.z, c0.z); h0[0][3] = __fsub_rn(x0.w, c0.w);
        h0[0][4] = __fsub_rn(x1.x, c1.x); h0[0][5] = __fsub_rn(x1.y, c1.y); h0[0][6] = __fsub_rn(x1.z, c1.z); h0[0][7] = __fsub_rn(x1.w, c1.w);
#pragma unroll
        for (int i = 8; i < 16; ++i) h0[0][i] = 0.f;
    }
    const int lw = wa + 4 * lane;                  // the lane's float4 within a k-group's 256 floats
    const int lb = wa + 4 * lh;                    // the lane's first channel within a block's group of 8
    f32x16 ha[2], hb[2];
    // ---- layer 0: K = 16 (two k-groups), 64 channels ----
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) { ha[0][i] = 0.f; ha[1][i] = 0.f; }
        float4 a0 = lrg_wb_lds4(LRG_WA_W0 + lw), a1 = lrg_wb_lds4(LRG_WA_W0 + 2 * 256 + lw);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float4 n0 = a0, n1 = a1;
            if (g == 0) { n0 = lrg_wb_lds4(LRG_WA_W0 + 256 + lw); n1 = lrg_wb_lds4(LRG_WA_W0 + 3 * 256 + lw); }
            const int r = 4 * g;
            ha[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, h0[0][r + 0], ha[0], 0, 0, 0);
            ha[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, h0[0][r + 0], ha[1], 0, 0, 0);
            ha[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, h0[0][r + 1], ha[0], 0, 0, 0);
            ha[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, h0[0][r + 1], ha[1], 0, 0, 0);
            ha[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, h0[0][r + 2], ha[0], 0, 0, 0);
            ha[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, h0[0][r + 2], ha[1], 0, 0, 0);
            ha[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, h0[0][r + 3], ha[0], 0, 0, 0);
            ha[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, h0[0][r + 3], ha[1], 0, 0, 0);
            a0 = n0; a1 = n1;
        }
        lrg_wb_bias_relu(ha[0], LRG_WA_B0 + lb);
        lrg_wb_bias_relu(ha[1], LRG_WA_B0 + 32 + lb);
    }
    // ---- layer 1: 64 -> 64, and its copy for the heads (conv[1], :130,:134) ----
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) { hb[0][i] = 0.f; hb[1][i] = 0.f; }
        lrg_wb_mfma2<8, 2>(hb[0], hb[1], ha, LRG_WA_W1 + lw, LRG_WA_W1 + 8 * 256 + lw);
        lrg_wb_bias_relu(hb[0], LRG_WA_B1 + lb);
        lrg_wb_bias_relu(hb[1], LRG_WA_B1 + 32 + lb);
        float *gb = conv1 + r0 * LRG_WB_C1;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                lrg_st_coh4(gb, (unsigned)(li * LRG_WB_C1 + 32 * b + 8 * j + 4 * lh) * 4u, make_float4(hb[b][4 * j], hb[b][4 * j + 1], hb[b][4 * j + 2], hb[b][4 * j + 3]));
    }
    // ---- layer 2: 64 -> 64 ----
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) { ha[0][i] = 0.f; ha[1][i] = 0.f; }
        lrg_wb_mfma2<8, 2>(ha[0], ha[1], hb, LRG_WA_W2 + lw, LRG_WA_W2 + 8 * 256 + lw);
        lrg_wb_bias_relu(ha[0], LRG_WA_B2 + lb);
        lrg_wb_bias_relu(ha[1], LRG_WA_B2 + 32 + lb);
    }
    // ---- layer 3: 64 -> 128, two blocks at a time, out to the tile's rows for the POOL tasks ----
    float *gh = h3out + r0 * LRG_WB_C3;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f32x16 c0, c1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
        lrg_wb_mfma2<8, 2>(c0, c1, ha, LRG_WA_W3 + (2 * p) * 8 * 256 + lw, LRG_WA_W3 + (2 * p + 1) * 8 * 256 + lw);
        lrg_wb_bias_relu(c0, LRG_WA_B3 + 64 * p + lb);
        lrg_wb_bias_relu(c1, LRG_WA_B3 + 64 * p + 32 + lb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lrg_st_coh4(gh, (unsigned)(li * LRG_WB_C3 + 64 * p + 8 * j + 4 * lh) * 4u, make_float4(c0[4 * j], c0[4 * j + 1], c0[4 * j + 2], c0[4 * j + 3]));
            lrg_st_coh4(gh, (unsigned)(li * LRG_WB_C3 + 64 * p + 32 + 8 * j + 4 * lh) * 4u, make_float4(c1[4 * j], c1[4 * j + 1], c1[4 * j + 2], c1[4 * j + 3]));
        }
    }
}

// POOL task: the tile's layer-3 rows h3in[r0 .. r0 + 31][128], quarter `q` of the pooled layer's columns (its kernel at LDS offset `wp`), block pairs p_lo .. p_hi - 1
// of the quarter's two; pool: the slot's pooled feature for this side ([512] floats, zero before the evaluation's first tile)
__device__ __forceinline__ void lrg_wave_pool_tile(const float *h3in, float *pool, long r0, int q, int wp, int p_lo, int p_hi, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    f32x16 h3[4];
    {
        const float *gh = h3in + r0 * LRG_WB_C3;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = lrg_ld_coh4(gh, (unsigned)(li * LRG_WB_C3 + 32 * b + 8 * j + 4 * lh) * 4u);
                h3[b][4 * j] = v.x; h3[b][4 * j + 1] = v.y; h3[b][4 * j + 2] = v.z; h3[b][4 * j + 3] = v.w;
            }
    }
    const int lw = wp + 4 * lane;
    const bool row1 = (lane & 16) != 0;
    for (int p = p_lo; p < p_hi; ++p) {
        f32x16 c0, c1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
        lrg_wb_mfma2<16, 4>(c0, c1, h3, LRG_WP_W4 + (2 * p) * 16 * 256 + lw, LRG_WP_W4 + (2 * p + 1) * 16 * 256 + lw);
        const float m0 = lrg_wb_rowmax16(c0, lane), m1 = lrg_wb_rowmax16(c1, lane);
        // rows 0 / 1 of a half hold the maxima over lanes 0-15 / 16-31 of the same channels: the even row keeps block 2p, the odd row block 2p + 1
        const float keep = row1 ? m1 : m0, send = row1 ? m0 : m1;
        const float m = fmaxf(keep, __shfl_xor(send, 16));
        const int t = lane & 15;
        const int ch = 32 * (2 * p + (row1 ? 1 : 0)) + 8 * (t >> 2) + 4 * lh + (t & 3);      // within the quarter
        const float v = fmaxf(m + LRG_WB_SMEM[wp + LRG_WP_B4 + ch], 0.f);
        const int bits = __float_as_int(v);
        if (bits > 0) atomicMax(reinterpret_cast<int *>(pool + q * 128 + ch), bits);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
// REGISTER TILE: the same 32-row branch tile by a TEAM of four wavefronts (one per SIMD, as lrg_fused_tile's), activations in registers where that is free.
// One wavefront per tile is one SIMD's matrix pipe per tile: 21-23 us through PREFIX and POOL against 17 (idle) / 25 us (loaded) for the team tile, whose passes
// spread over four pipes (profiles/r06_wave_sweep_*.txt).  The team tile's time, by its cycle stamps (profiles/r04_delay_and_stamps.txt: 52 k cycles for 23 k of MFMA
// issue), is barriers, LDS round trips and epilogues of the NARROW layers: 25 k cycles before the pooled layer starts, for 272 MFMAs of which every wave issues a
// quarter or a half.  Here:
//   layers 0 - 2 (13 -> 64 -> 64 -> 64, 144 MFMAs): every wavefront computes them for ITSELF, in registers (kernels from LDS) -- 9 k cycles of matrix pipe on four
//     pipes at once, no barrier, no LDS round trip, instead of three passes of ~5 k cycles each with two waves idle;
//   layer 3 (64 -> 128): wavefront w computes column block w (its kernel columns prefetched from L2 at the task's start), the four blocks meet in LDS -- the tile's ONE
//     barrier -- and every wavefront takes all 128 channels back as its B operands;
//   layer 4 (128 -> 512, pooled): wavefront w computes column blocks 4w .. 4w + 3, the kernel columns requested from L2 one whole block (sixteen k-groups) ahead;
//     the max over the points by the transposing DPP reduction, one atomicMax instruction per pair of blocks.
// ~190 VGPRs: a tile of lrg_grow_async_worker_kernel (512 threads) only.  Same sums in the same order: bit-identical to the team tile.
#define LRG_RT_W0 0                    // per side: layer 0 (1024 floats), 1, 2 (4096 each); biases of layers 0 - 4 (64, 64, 64, 128, 512)
#define LRG_RT_W1 1024
#define LRG_RT_W2 5120
#define LRG_RT_B0 9216
#define LRG_RT_B1 9280
#define LRG_RT_B2 9344
#define LRG_RT_B3 9408
#define LRG_RT_B4 9536
#define LRG_RT_SIDE 10048
#define LRG_RT_WEIGHT_FLOATS (2 * LRG_RT_SIDE)
#define LRG_RT_XCH_LD 132              // floats between two points' rows of the layer-3 exchange (128 + 4: conflict-free 16-byte accesses)
#define LRG_RT_XCH_FLOATS (32 * LRG_RT_XCH_LD)

// 16 bytes of a kernel image through a buffer descriptor on a wave-uniform base: the lane's offset in ONE register, the k-group's in a scalar -- as global loads every
// one of a tile's 72 kernel loads had a 64-bit address of its own in registers (offsets beyond the instruction's 4 KB immediate), and the tile spilled
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float4 lrg_rt_ldw(const __amdgpu_buffer_rsrc_t &r, unsigned lane_off, unsigned group_off) {
    const lrg_u32x4v u = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, group_off, 0);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
#else
struct lrg_rt_host_rsrc {};
#define __amdgpu_buffer_rsrc_t lrg_rt_host_rsrc
#define __builtin_amdgcn_make_buffer_rsrc(...) lrg_rt_host_rsrc()
__device__ __forceinline__ float4 lrg_rt_ldw(const lrg_rt_host_rsrc &, unsigned, unsigned) { return make_float4(0.f, 0.f, 0.f, 0.f); }
#endif
// one 32-channel block over NG k-groups, kernel operands from registers (a[g]), activations h (NG / 4 blocks)
template <int NG, int NBI>
__device__ __forceinline__ void lrg_rt_mfma1(f32x16 &acc, const f32x16 (&h)[NBI], const float4 (&a)[NG]) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const f32x16 &hb = h[g >> 2];
        const int r = 4 * (g & 3);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, hb[r + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, hb[r + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, hb[r + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, hb[r + 3], acc, 0, 0, 0);
    }
}

// wn: the wavefront's number in its team (0 .. 3); ws: LDS offset (floats) of the side's kernels (LRG_RT_*); xch: LDS offset of the team's exchange buffer;
// w3 / w4: the packed images of layers 3 and 4 in global memory (lrg_pack_weights layout)
// part / nparts (1 or 2): the tile as two tasks on two teams that each run layers 0 - 3 and HALF of every wavefront's column blocks of the pooled layer (blocks
// 4 wn + 2 part, + 1): where CUs idle (few slots in flight) the pooled layer's 20 k cycles are what a tile's latency is made of; part 0 alone stores conv[1].
template <int NPARTS, class TEAM>
__device__ __forceinline__ void lrg_team_branch_tile_reg(const float *x, const float *center, float *conv1, float *pool, const float *w3, const float *w4, long r0, int slot,
                                                         int ws, int xch, const TEAM &team, int wn, int lane, int part = 0) {
    static_assert(NPARTS == 1 || NPARTS == 2, "one or two tasks per tile");
    const int li = lane & 31, lh = lane >> 5;
    // ---- what comes from memory, requested first: the rows, the centre, this wavefront's columns of layer 3 and its first block of layer 4 ----
    const float4 x0 = lrg_ld_coh4(x + r0 * 16, (unsigned)(li * 16 + 4 * lh) * 4u), x1 = lrg_ld_coh4(x + r0 * 16, (unsigned)(li * 16 + 8 + 4 * lh) * 4u);
    const float4 c0 = lrg_ld_coh4(center + (long)slot * 16, (unsigned)(4 * lh) * 4u), c1 = lrg_ld_coh4(center + (long)slot * 16, (unsigned)(8 + 4 * lh) * 4u);
    float4 a3[8], wa[16];
    // (block wn of layer 3: 8 k-groups of 1 KB; blocks 4 wn .. 4 wn + 3 of layer 4: 16 k-groups each)
    const __amdgpu_buffer_rsrc_t r3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w3) + (long)(wn * 8) * 256, 0, 0x7fffffff, 0x00020000);
    const int q_lo = NPARTS == 2 ? 2 * part : 0;
    constexpr int q_n = NPARTS == 2 ? 2 : 4;      // this task's column blocks of the wavefront's four
    const __amdgpu_buffer_rsrc_t r4 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w4) + (long)((4 * wn + q_lo) * 16) * 256, 0, 0x7fffffff, 0x00020000);
    const unsigned lo = (unsigned)lane * 16u;
#pragma unroll
    for (int g = 0; g < 8; ++g) a3[g] = lrg_rt_ldw(r3, lo, (unsigned)g * 1024u);
    f32x16 h0[1];
    h0[0][0] = __fsub_rn(x0.x, c0.x); h0[0][1] = __fsub_rn(x0.y, c0.y); h0[0][2] = __fsub_rn(x0.z, c0.z); h0[0][3] = __fsub_rn(x0.w, c0.w);
    h0[0][4] = __fsub_rn(x1.x, c1.x); h0[0][5] = __fsub_rn(x1.y, c1.y); h0[0][6] = __fsub_rn(x1.z, c1.z); h0[0][7] = __fsub_rn(x1.w, c1.w);
#pragma unroll
    for (int i = 8; i < 16; ++i) h0[0][i] = 0.f;
    const int lw = ws + 4 * lane, lb = ws + 4 * lh;
    f32x16 ha[2], hb[2];
    // ---- layer 0 ----
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) { ha[0][i] = 0.f; ha[1][i] = 0.f; }
        float4 a0 = lrg_wb_lds4(LRG_RT_W0 + lw), a1 = lrg_wb_lds4(LRG_RT_W0 + 2 * 256 + lw);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float4 n0 = a0, n1 = a1;
            if (g == 0) { n0 = lrg_wb_lds4(LRG_RT_W0 + 256 + lw); n1 = lrg_wb_lds4(LRG_RT_W0 + 3 * 256 + lw); }
            const int r = 4 * g;
            ha[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, h0[0][r + 0], ha[0], 0, 0, 0);
            ha[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, h0[0][r + 0], ha[1], 0, 0, 0);
            ha[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, h0[0][r + 1], ha[0], 0, 0, 0);
            ha[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, h0[0][r + 1], ha[1], 0, 0, 0);
            ha[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, h0[0][r + 2], ha[0], 0, 0, 0);
            ha[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, h0[0][r + 2], ha[1], 0, 0, 0);
            ha[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, h0[0][r + 3], ha[0], 0, 0, 0);
            ha[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, h0[0][r + 3], ha[1], 0, 0, 0);
            a0 = n0; a1 = n1;
        }
        lrg_wb_bias_relu(ha[0], LRG_RT_B0 + lb);
        lrg_wb_bias_relu(ha[1], LRG_RT_B0 + 32 + lb);
    }
    // ---- layer 1 (+ conv[1] for the heads, by the team's first wavefront) ----
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) { hb[0][i] = 0.f; hb[1][i] = 0.f; }
        lrg_wb_mfma2<8, 2>(hb[0], hb[1], ha, LRG_RT_W1 + lw, LRG_RT_W1 + 8 * 256 + lw);
        lrg_wb_bias_relu(hb[0], LRG_RT_B1 + lb);
        lrg_wb_bias_relu(hb[1], LRG_RT_B1 + 32 + lb);
        if (wn == 0 && part == 0) {
            float *gb = conv1 + r0 * LRG_WB_C1;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    lrg_st_coh4(gb, (unsigned)(li * LRG_WB_C1 + 32 * b + 8 * j + 4 * lh) * 4u, make_float4(hb[b][4 * j], hb[b][4 * j + 1], hb[b][4 * j + 2], hb[b][4 * j + 3]));
        }
    }
    // ---- layer 2 ----
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) { ha[0][i] = 0.f; ha[1][i] = 0.f; }
        lrg_wb_mfma2<8, 2>(ha[0], ha[1], hb, LRG_RT_W2 + lw, LRG_RT_W2 + 8 * 256 + lw);
        lrg_wb_bias_relu(ha[0], LRG_RT_B2 + lb);
        lrg_wb_bias_relu(ha[1], LRG_RT_B2 + 32 + lb);
    }
    // ---- layer 3: this wavefront's column block, then all four through LDS ----
    // (the first block of layer 4's kernel columns starts its trip here: behind layers 0 - 2, whose 96 live registers it would have sat beside, and ahead of this
    //  layer's 32 MFMAs, the exchange and the barrier -- ~4 k cycles)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 16; ++g) wa[g] = lrg_rt_ldw(r4, lo, (unsigned)g * 1024u);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 h3[4];
    {
        f32x16 c;
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
        lrg_rt_mfma1<8, 2>(c, ha, a3);
        lrg_wb_bias_relu(c, LRG_RT_B3 + 32 * wn + lb);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4 *>(&LRG_WB_SMEM[xch + li * LRG_RT_XCH_LD + 32 * wn + 8 * j + 4 * lh]) = make_float4(c[4 * j], c[4 * j + 1], c[4 * j + 2], c[4 * j + 3]);
    }
    team.sync();
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = lrg_wb_lds4(xch + li * LRG_RT_XCH_LD + 32 * b + 8 * j + 4 * lh);
            h3[b][4 * j] = v.x; h3[b][4 * j + 1] = v.y; h3[b][4 * j + 2] = v.z; h3[b][4 * j + 3] = v.w;
        }
    // ---- layer 4: column blocks 4 wn .. 4 wn + 3; a k-group's kernel operand is replaced by the NEXT block's right behind its last MFMA (sixteen k-groups = 4 k
    //      cycles of lead over a trip to L2: one load per four MFMAs in the stream, 64 registers instead of the 128 of two whole blocks -- which spilled) ----
    const bool row1 = (lane & 16) != 0;
    const int t = lane & 15;
    float m_even = 0.f;
#pragma unroll
    for (int q = 0; q < q_n; ++q) {
        f32x16 c;
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const f32x16 &hb4 = h3[g >> 2];
            const int r = 4 * (g & 3);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[g].x, hb4[r + 0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[g].y, hb4[r + 1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[g].z, hb4[r + 2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[g].w, hb4[r + 3], c, 0, 0, 0);
            if (q + 1 < q_n) wa[g] = lrg_rt_ldw(r4, lo, (unsigned)((q + 1) * 16 + g) * 1024u);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float m = lrg_wb_rowmax16(c, lane);
        if (!(q & 1)) m_even = m;
        else {
            // rows 0 / 1 of a half hold the maxima over lanes 0-15 / 16-31 of the same channels: the even row keeps block q - 1, the odd row block q
            const float keep = row1 ? m : m_even, send = row1 ? m_even : m;
            const float mm = fmaxf(keep, __shfl_xor(send, 16));
            const int ch = 32 * (4 * wn + q_lo + q - 1 + (row1 ? 1 : 0)) + 8 * (t >> 2) + 4 * lh + (t & 3);
            const float v = fmaxf(mm + LRG_WB_SMEM[ws + LRG_RT_B4 + ch], 0.f);
            const int bits = __float_as_int(v);
            if (bits > 0) atomicMax(reinterpret_cast<int *>(pool + ch), bits);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
// REGISTER HEAD TILE: a 32-row tile of a head stack (learn_region_grow_util.py:138-162: conv[1] (64) + the hoisted pooled product -> 256 -> 128 -> 2) by a team of four
// wavefronts.  The team tile (lrg_fused_tile) stages the rows in LDS, runs the 64 -> 256 layer as two passes of which only the FIRST one's MFMAs run before it waits for
// the slot's pooled product, and meets at four barriers: 13 us from the pooled product to the logits (profiles/r06_bench_debug_68_update_by_lists.log).  Here:
//   conv[1] rows straight into registers as B operands; layer 0: wavefront w computes column blocks 2w, 2w + 1 (kernel columns from L2, requested with the rows) --
//     ALL of the layer's MFMAs before the wait; behind it only the pooled product's 8 values per lane, ReLU and the exchange (LDS, barrier);
//   layer 1 (256 -> 128): wavefront w computes column block w, B operands from the exchange buffer two k-groups ahead, kernel columns from L2 sixteen k-groups ahead;
//   the 2-wide last layer from a second buffer in LDS: lrg_fused_tile's own arithmetic (eight lanes per row, a fixed butterfly), so the same bits.
// Same sums in the same order as the team tile: bit-identical logits.
#define LRG_RH_H0_LD 260
#define LRG_RH_H1_LD 132
#define LRG_RH_H0 0                                  // [32][260] layer 0's output
#define LRG_RH_H1 (32 * LRG_RH_H0_LD)                // [32][132] layer 1's output
#define LRG_RH_FW (LRG_RH_H1 + 32 * LRG_RH_H1_LD)    // [128][2] the last layer's kernel
#define LRG_RH_FLOATS (LRG_RH_FW + 256)

template <class TEAM, class WAIT>
__device__ __forceinline__ void lrg_team_head_tile_reg(const LrgFusedProb &P, long r0, int slot, int sm, const TEAM &team, const WAIT &wait_pooled, int wn, int lane) {
    const int li = lane & 31, lh = lane >> 5, tid = 64 * wn + lane;
    const unsigned lo = (unsigned)lane * 16u;
    // ---- requested first: the conv[1] rows, this wavefront's two column blocks of layer 0, its first sixteen k-groups of layer 1, layer 1's bias ----
    f32x16 hin[2];
    {
        const float *gx = P.x + r0 * 64;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = lrg_ld_coh4(gx, (unsigned)(li * 64 + 32 * b + 8 * j + 4 * lh) * 4u);
                hin[b][4 * j] = v.x; hin[b][4 * j + 1] = v.y; hin[b][4 * j + 2] = v.z; hin[b][4 * j + 3] = v.w;
            }
    }
    const __amdgpu_buffer_rsrc_t q0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.L[0].w) + (long)(2 * wn * 8) * 256, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t q1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.L[1].w) + (long)(wn * 32) * 256, 0, 0x7fffffff, 0x00020000);
    float4 a0[8], a1[8], wr[16], b1[4];
#pragma unroll
    for (int g = 0; g < 8; ++g) { a0[g] = lrg_rt_ldw(q0, lo, (unsigned)g * 1024u); a1[g] = lrg_rt_ldw(q0, lo, (unsigned)(8 + g) * 1024u); }
#pragma unroll
    for (int g = 0; g < 16; ++g) wr[g] = lrg_rt_ldw(q1, lo, (unsigned)g * 1024u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float *bp1 = P.L[1].bias + 32 * wn + 8 * j + 4 * lh;
        b1[j] = make_float4(bp1[0], bp1[1], bp1[2], bp1[3]);
    }
    LRG_WB_SMEM[sm + LRG_RH_FW + tid] = P.fw[tid];                     // (256 floats: one per thread of the team)
    // ---- layer 0: 64 -> this wavefront's 64 of 256 columns, all of it before the pooled product is needed ----
    f32x16 c0, c1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const f32x16 &hb = hin[g >> 2];
        const int r = 4 * (g & 3);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[g].x, hb[r + 0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[g].x, hb[r + 0], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[g].y, hb[r + 1], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[g].y, hb[r + 1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[g].z, hb[r + 2], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[g].z, hb[r + 2], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[g].w, hb[r + 3], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[g].w, hb[r + 3], c1, 0, 0, 0);
    }
    // ---- the slot's pooled product (the hoisted part of layer 0, :128-141) as a per-slot bias, then ReLU; the blocks meet in LDS ----
    wait_pooled();
    {
        const float *hbp = P.L[0].bias + (long)slot * 256;
        float4 hv[8];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[4 * b + j] = lrg_ld_coh4(hbp, (unsigned)(32 * (2 * wn + b) + 8 * j + 4 * lh) * 4u);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x16 &c = b ? c1 : c0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (the team tile adds the per-slot bias, then its -- zero -- layer bias, then takes the ReLU: the same three operations)
                const float4 h = hv[4 * b + j];
                const float v0 = fmaxf(__fadd_rn(__fadd_rn(c[4 * j + 0], h.x), 0.f), 0.f), v1 = fmaxf(__fadd_rn(__fadd_rn(c[4 * j + 1], h.y), 0.f), 0.f);
                const float v2 = fmaxf(__fadd_rn(__fadd_rn(c[4 * j + 2], h.z), 0.f), 0.f), v3 = fmaxf(__fadd_rn(__fadd_rn(c[4 * j + 3], h.w), 0.f), 0.f);
                *reinterpret_cast<float4 *>(&LRG_WB_SMEM[sm + LRG_RH_H0 + li * LRG_RH_H0_LD + 32 * (2 * wn + b) + 8 * j + 4 * lh]) = make_float4(v0, v1, v2, v3);
            }
        }
    }
    team.sync();
    // ---- layer 1: 256 -> this wavefront's 32 of 128 columns ----
    {
        f32x16 c;
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
        const int bp = sm + LRG_RH_H0 + li * LRG_RH_H0_LD + 4 * lh;
        float4 br[3];
        br[0] = lrg_wb_lds4(bp); br[1] = lrg_wb_lds4(bp + 8);
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            if (g + 2 < 32) br[(g + 2) % 3] = lrg_wb_lds4(bp + 8 * (g + 2));
            const float4 a = wr[g & 15], bq = br[g % 3];
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, c, 0, 0, 0);
            if (g + 16 < 32) wr[g & 15] = lrg_rt_ldw(q1, lo, (unsigned)(g + 16) * 1024u);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4 *>(&LRG_WB_SMEM[sm + LRG_RH_H1 + li * LRG_RH_H1_LD + 32 * wn + 8 * j + 4 * lh]) =
                make_float4(fmaxf(c[4 * j + 0] + b1[j].x, 0.f), fmaxf(c[4 * j + 1] + b1[j].y, 0.f), fmaxf(c[4 * j + 2] + b1[j].z, 0.f), fmaxf(c[4 * j + 3] + b1[j].w, 0.f));
    }
    team.sync();
    // ---- the 2-wide last layer, no ReLU (:145-149, :158-162): lrg_fused_tile's arithmetic -- eight lanes per row, each every eighth float4 of the row, the partial
    //      sums combined by xor-shuffles in a fixed order ----
    {
        const float *act = &LRG_WB_SMEM[sm + LRG_RH_H1], *fw = &LRG_WB_SMEM[sm + LRG_RH_FW];
        const int row = tid >> 3, q = tid & 7;
        float s0 = 0.f, s1 = 0.f;
        for (int k = 4 * q; k < 128; k += 32) {
            const float4 a = *reinterpret_cast<const float4 *>(act + row * LRG_RH_H1_LD + k);
            const float4 w01 = *reinterpret_cast<const float4 *>(fw + 2 * k);
            const float4 w23 = *reinterpret_cast<const float4 *>(fw + 2 * k + 4);
            s0 = fmaf(a.x, w01.x, s0); s1 = fmaf(a.x, w01.y, s1);
            s0 = fmaf(a.y, w01.z, s0); s1 = fmaf(a.y, w01.w, s1);
            s0 = fmaf(a.z, w23.x, s0); s1 = fmaf(a.z, w23.y, s1);
            s0 = fmaf(a.w, w23.z, s0); s1 = fmaf(a.w, w23.w, s1);
        }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) { s0 += __shfl_xor(s0, m); s1 += __shfl_xor(s1, m); }
        // two rows' logits per 16-byte write-through store
        const float t0 = s0 + P.fb[0], t1 = s1 + P.fb[1];
        const float u0 = __shfl_down(t0, 8), u1 = __shfl_down(t1, 8);
        if (q == 0 && !(row & 1)) lrg_st_coh4(P.fout + r0 * 2, (unsigned)row * 8u, make_float4(t0, t1, u0, u1));
    }
}
