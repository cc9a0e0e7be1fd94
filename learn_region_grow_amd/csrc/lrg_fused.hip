// Fused LrgNet stacks for gfx950: a whole branch (learn_region_grow_util.py:106-123) or a whole head (:138-162)
// per 64-row tile in ONE kernel.  Activations never leave the CU: they ping-pong between two LDS buffers; the layer
// weights (L2-resident, 3.2 MB in all) go straight into the MFMA B operands, a few k-groups ahead of the fp32 MFMAs
// (v_mfma_f32_32x32x2_f32, exact fp32).  Only conv[1] (needed by the heads), the pooled maxima and the logits are
// written to HBM; the 512-wide layer and the 1088-wide concat are never materialised.
//
// The layer-streamed formulation (one launch per layer, lrg_net.hip) stays available: it is what the layer-by-layer
// parity tests and the "HBM-streamed" roofline figure use.
#include "lrg_common.h"
#include "lrg_fused.h"

#ifndef LRG_TRACE
#define LRG_TRACE 0     // 1: thread 0 of every workgroup stamps s_memtime at phase boundaries into P.fout-adjacent debug memory
#endif
#if LRG_TRACE
__device__ long long *g_lrg_trace = nullptr;
extern "C" void lrg_set_trace(long long *p) { hipMemcpyToSymbol(HIP_SYMBOL(g_lrg_trace), &p, sizeof(p)); }
#define TRACE(i) do { if (tid == 0 && g_lrg_trace && blockIdx.x < 2048) g_lrg_trace[((long)blockIdx.y * 2048 + blockIdx.x) * 32 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TRACE(i)
#endif
#ifndef LRG_ABLATE
#define LRG_ABLATE 0   // timing experiments only: 1 = no weight loads in the MFMA loop, 2 = no LDS reads (results wrong)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FBN 128      // output columns per pass: 4 waves side by side, each a 64x32 strip (two 32x32 MFMA tiles sharing B)
#define FTHREADS 256 // one wave per SIMD per workgroup; 2-3 workgroups per CU interleave without sharing barriers
#define FDIST 4      // k-groups of B operands in flight ahead of the MFMAs (4 groups = 32 MFMAs = 2048 cycles)

// A (32*RT)x32 output strip (RT 32x32 tiles stacked in rows) over NG k-groups of 8.  Lane half h feeds logical
// k = 8g + 4h + s of both operands:
//   A (activations) from LDS, one ds_read_b128 per tile per group;
//   B (weights) straight from L2/L1 into registers: four coalesced global_load_dword per group (each fetches two
//   128-B lines: 32 consecutive columns of rows k and k+4), shared by the RT tiles.  No wave shares its B columns
//   with another wave, so an LDS round trip would buy nothing and its barriers would serialise the waves.
//   wrow = wave-uniform base (SGPRs), loff = per-lane element offset: saddr-form loads, no 64-bit VALU address math.
template <int NG, int RT>
__device__ __forceinline__ void tile_mfma(f32x16 (&acc)[RT], const float *ap, int ld_in, const float *wrow, int loff,
                                          int ldw) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float4 a[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) a[t] = *reinterpret_cast<const float4 *>(ap + t * 32 * ld_in + 8 * g);
        const float *w0 = wrow + (long)(8 * g) * ldw;
        float b0 = w0[loff];
        float b1 = w0[loff + ldw];
        float b2 = w0[loff + 2 * ldw];
        float b3 = w0[loff + 3 * ldw];
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b0, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b1, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b2, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b3, acc[t], 0, 0, 0);
    }
    // issue order: FDIST groups of weight loads and 2 groups of LDS reads run ahead of the MFMAs
    constexpr int D = NG < FDIST ? NG : FDIST;
    constexpr int DA = NG < 2 ? NG : 2;
    __builtin_amdgcn_sched_group_barrier(0x020, 4 * D, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, RT * DA, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0);
        if (g + D < NG) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
        if (g + DA < NG) __builtin_amdgcn_sched_group_barrier(0x100, RT, 0);
    }
}

// Ragged K (the 13-wide first layer, padded to 16 in LDS): rows >= K of the weights read as zero.
template <int RT>
__device__ __forceinline__ void tile_mfma_ragged(f32x16 (&acc)[RT], const float *ap, int ld_in, const float *wrow, int loff,
                                                 int ldw, int K, int kbase) {
    const int ng = (K + 7) >> 3;
    for (int g = 0; g < ng; ++g) {
        float4 a[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) a[t] = *reinterpret_cast<const float4 *>(ap + t * 32 * ld_in + 8 * g);
        float b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) b[s] = (kbase + 8 * g + s < K) ? wrow[(long)(8 * g + s) * ldw + loff] : 0.f;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[3], acc[t], 0, 0, 0);
        }
    }
}

template <int CAP0, int CAP1, int RT>
__global__ __launch_bounds__(FTHREADS) void lrg_fused_stack_kernel(LrgFusedArgs args) {
    constexpr int FM = 32 * RT;      // rows (points) per workgroup
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *buf0 = smem;                       // outputs of even layers
    float *buf1 = smem + CAP0;                // the staged input and outputs of odd layers
    float *poolbuf = smem + CAP0 + CAP1;      // [512] running column maxima of the pooled layer / final-layer weights

    const LrgFusedProb &P = args.p[blockIdx.y];
    // Tile-major block order (instance fastest): block b runs on XCD b % 8, and with duplicate-row skipping mostly the
    // FIRST tiles of the instances survive -- instance-major order would put all of them on one XCD.
    const int ninst = (int)(P.rows / P.rows_per_inst);
    const int inst = blockIdx.x % ninst, tile = blockIdx.x / ninst;
    if (tile * FM >= P.rows_per_inst) return;
    // rows beyond valid[instance] are copies of earlier rows (the padding rule, test_region_grow.py:240,:252):
    // their per-point results are identical and the max-pool ignores duplicates, so whole tiles of them are skipped
    if (P.valid && tile * FM >= P.valid[inst]) return;
    const long r0 = (long)inst * P.rows_per_inst + (long)tile * FM;
    // Column split of the pooled (last, widest) layer over gridDim.z workgroups: each recomputes the narrow layers and
    // takes every gridDim.z-th 128-column block of the last one.  It shortens the critical path of a tile ~2.5x when
    // few tiles are live (duplicate-row skipping) at the price of ~1.5x the MFMA work, so it is dropped -- z > 0
    // workgroups exit, z = 0 does everything -- once the live-tile count says the chip would be full anyway.
    int zsplit = gridDim.z;
    if (zsplit > 1 && P.tile_total && *P.tile_total * zsplit > P.split_limit) zsplit = 1;
    const int zme = blockIdx.z;
    if (zme >= zsplit) return;
    if (zsplit > 1 && zme >= (P.L[P.nlayers - 1].N + FBN - 1) / FBN) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    TRACE(0);

    // ---- stage the input rows into buf1, zero-padded to a multiple of 8 columns ----
    const int Kin = P.Kin;
    const int Kp = (Kin + 7) & ~7;
    const int ld_x = Kp + 4;
    if ((P.ldx & 3) == 0 && (Kin & 3) == 0 && (((uintptr_t)P.x) & 15) == 0) {
        const int q = Kp >> 2;
        for (int idx = tid; idx < FM * q; idx += FTHREADS) {
            int row = idx / q, c4 = idx - row * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * c4 < Kin) v = *reinterpret_cast<const float4 *>(P.x + (r0 + row) * P.ldx + 4 * c4);
            *reinterpret_cast<float4 *>(&buf1[row * ld_x + 4 * c4]) = v;
        }
    } else {
        for (int idx = tid; idx < FM * Kp; idx += FTHREADS) {
            int row = idx / Kp, c = idx - row * Kp;
            buf1[row * ld_x + c] = c < Kin ? P.x[(r0 + row) * P.ldx + c] : 0.f;
        }
    }
    for (int i = tid; i < 512; i += FTHREADS) poolbuf[i] = 0.f;
    __syncthreads();
    TRACE(1);

    const int nlayers = P.nlayers;
    int prevN = Kp;
    int lastN = 0, lastflags = 0;
    LrgFusedLayer Lnext = P.L[0];
    for (int l = 0; l < nlayers; ++l) {
        const LrgFusedLayer L = Lnext;             // descriptors are fetched one layer ahead (scalar loads off the critical path)
        if (l + 1 < nlayers) Lnext = P.L[l + 1];
        const bool inplace = (L.flags & LRG_FL_INPLACE) != 0;
        const float *act_in = (l & 1) ? buf0 : buf1;
        float *act_out = ((l & 1) != 0) == !inplace ? buf1 : buf0;
        const int ld_in = prevN + 4, ld_out = L.N + 4;
        const float *ap = act_in + li * ld_in + 4 * lh;
        const int loff = 4 * lh * L.ldw + li;
        const int ncb = (L.N + FBN - 1) / FBN;
        const bool split_here = zsplit > 1 && l == nlayers - 1;
        TRACE(2 + 2 * l);
        for (int cb = split_here ? zme : 0; cb < ncb; cb += split_here ? zsplit : 1) {
            const int col0 = cb * FBN + wn * 32;
            const bool wave_on = col0 < L.N;         // 64-wide layers keep only two of the four waves busy
            f32x16 acc[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
            float bv = 0.f;                          // issued before the MFMAs so that its latency hides behind them
            if (wave_on && L.bias)
                bv = (L.flags & LRG_FL_INST_BIAS) ? L.bias[(r0 / P.rows_per_inst) * L.N + col0 + li] : L.bias[col0 + li];
            if (wave_on) {
                const float *wrow = L.w + col0;
                if (L.K == 128) tile_mfma<16, RT>(acc, ap, ld_in, wrow, loff, L.ldw);
                else if (L.K == 64) tile_mfma<8, RT>(acc, ap, ld_in, wrow, loff, L.ldw);
                else if (L.K == 256) tile_mfma<32, RT>(acc, ap, ld_in, wrow, loff, L.ldw);
                else tile_mfma_ragged<RT>(acc, ap, ld_in, wrow, loff, L.ldw, L.K, 4 * lh);
            }
            if (inplace) __syncthreads();            // the output overlays this layer's input: everyone must be done reading
            if (wave_on) {
                // ---- epilogue: bias, ReLU, keep in LDS / copy to HBM / column max ----
                const int col = col0 + li;
                float cmax = 0.f;
#pragma unroll
                for (int r = 0; r < 16 * RT; ++r) {
                    const int rr = r & 15;
                    const int rl = (r >> 4) * 32 + 4 * lh + (rr & 3) + 8 * (rr >> 2);
                    float v = acc[r >> 4][rr] + bv;
                    if (L.flags & LRG_FL_RELU) v = fmaxf(v, 0.f);
                    if (L.flags & LRG_FL_KEEP) act_out[rl * ld_out + col] = v;
                    if (L.gout && (zme == 0 || split_here)) L.gout[(r0 + rl) * L.N + col] = v;
                    cmax = fmaxf(cmax, v);
                }
                if (L.flags & LRG_FL_POOL) {
                    cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
                    if (lh == 0) poolbuf[col] = fmaxf(poolbuf[col], cmax);   // this wave owns the column: no atomic needed
                }
            }
        }
        __syncthreads();                             // layer boundary: outputs visible, inputs dead
        TRACE(2 + 2 * l + 1);
        prevN = L.N;
        lastN = L.N;
        lastflags = L.flags;
    }

    // ---- pooled maxima of this tile -> the instance's pooled feature (:122-125) ----
    if ((lastflags & LRG_FL_POOL) && P.pool) {
        float *dst = P.pool + (r0 / P.rows_per_inst) * P.pool_stride;
        for (int c = tid; c < lastN; c += FTHREADS) atomicMax(reinterpret_cast<int *>(&dst[c]), __float_as_int(poolbuf[c]));
    }
    // ---- final 2-wide layer of a head, no ReLU (:145-149, :158-162) ----
    if (P.fw && zme == 0) {
        const int C = lastN;
        const bool odd = ((nlayers - 1) & 1) != 0;
        const float *act = (odd == !(lastflags & LRG_FL_INPLACE)) ? buf1 : buf0;
        const int ld = C + 4;
        for (int i = tid; i < 2 * C; i += FTHREADS) poolbuf[i] = P.fw[i];      // C <= 256: fits the 512-float scratch
        __syncthreads();
        // 4 lanes per row, interleaved k, combined by two xor-shuffles (fixed order: deterministic)
        const int row = tid >> 2, q = tid & 3;
        float s0 = 0.f, s1 = 0.f;
        if (row < FM)
            for (int k = q; k < C; k += 4) {
                float a = act[row * ld + k];
                s0 = fmaf(a, poolbuf[2 * k], s0);
                s1 = fmaf(a, poolbuf[2 * k + 1], s1);
            }
        s0 += __shfl_xor(s0, 1); s1 += __shfl_xor(s1, 1);
        s0 += __shfl_xor(s0, 2); s1 += __shfl_xor(s1, 2);
        if (row < FM && q == 0) {
            P.fout[(r0 + row) * 2 + 0] = s0 + P.fb[0];
            P.fout[(r0 + row) * 2 + 1] = s1 + P.fb[1];
        }
    }
    // ---- leave the pooled feature of this instance zero for the next evaluation (it was consumed by the GEMV) ----
    if (P.zero_pool && tile == 0 && zme == 0)
        for (int c = tid; c < P.zero_count; c += FTHREADS) P.zero_pool[(long)inst * P.zero_count + c] = 0.f;
    TRACE(20);
}

template <int CAP0, int CAP1, int RT>
static int launch_stack(const LrgFusedArgs &a, int nprob, int split, hipStream_t st) {
    constexpr int FM = 32 * RT;
    long maxrows = 0;
    for (int i = 0; i < nprob; ++i) {
        const LrgFusedProb &P = a.p[i];
        if (P.rows % FM != 0 || P.rows_per_inst % FM != 0) return LRG_EINVAL - 30;
        if (P.nlayers < 1 || P.nlayers > LRG_FUSED_MAXL) return LRG_EINVAL - 31;
        const int Kp = (P.Kin + 7) & ~7;
        if (FM * (Kp + 4) > CAP1) return LRG_EINVAL - 32;
        for (int l = 0; l < P.nlayers; ++l) {
            const LrgFusedLayer &L = P.L[l];
            if (L.N % 64 != 0 || L.N > 512 || L.ldw < L.N) return LRG_EINVAL - 33;
            if (l > 0 && (L.K != P.L[l - 1].N || (L.K != 64 && L.K != 128 && L.K != 256))) return LRG_EINVAL - 34;
            if (l == 0 && L.K != P.Kin) return LRG_EINVAL - 35;
            const bool inplace = (L.flags & LRG_FL_INPLACE) != 0;
            const bool to_buf1 = ((l & 1) != 0) == !inplace;
            if ((L.flags & LRG_FL_KEEP) && FM * (L.N + 4) > (to_buf1 ? CAP1 : CAP0)) return LRG_EINVAL - 36;
            if (inplace && (L.N > FBN || l + 1 != P.nlayers)) return LRG_EINVAL - 38;   // single column block, last layer only
            if (l + 1 < P.nlayers && !(L.flags & LRG_FL_KEEP)) return LRG_EINVAL - 37;
        }
        if (P.fw && P.L[P.nlayers - 1].N > 256) return LRG_EINVAL - 39;
        if (P.rows > maxrows) maxrows = P.rows;
    }
    if (maxrows == 0) return 0;
    const size_t lds = (size_t)(CAP0 + CAP1 + 512) * sizeof(float);
    auto kern = lrg_fused_stack_kernel<CAP0, CAP1, RT>;
    static bool attr_done = false;      // raising the dynamic-LDS cap is idempotent
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(maxrows / FM), nprob, split < 1 ? 1 : split), dim3(FTHREADS), lds, st, a);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_fused_branches(const LrgFusedArgs &a, int nprob, int split, hipStream_t st) {
    return launch_stack<64 * 68, 64 * 132, 2>(a, nprob, split, st);       // lite 0/1/2: hidden widths 64 / 128
}

int lrg_fused_heads(const LrgFusedArgs &a, int nprob, hipStream_t st) {
    return launch_stack<32 * 260, 32 * 68, 1>(a, nprob, 1, st);      // 64 -> 256 -> 128 (-> 2); the last hidden layer is written in place
}
