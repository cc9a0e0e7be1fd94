// LrgNet forward on gfx950: per-point layers on the fp32 matrix cores, max-pool, hoisted heads.
//
// Reference graph: learn_region_grow_util.py:106-162 (two 5-layer per-point branches, max-pool over the
// 512 rows of each, [pooled | conv[1]] concat, two 3-layer heads).  Restated here as
//   lrg_pointwise_mfma_kernel  y = relu(x @ W + b)  -- v_mfma_f32_32x32x2_f32 (exact fp32, == fmaf chain)
//   lrg_segmax_kernel          column max over an instance's rows
//   lrg_head_gemv_kernel /     pooled @ W0[:2*C_last] + b0  once per instance (the tiled concat is never built)
//   lrg_head_gemm_kernel       (the same on the matrix cores for batches of 8+ instances)
//   lrg_head_final_kernel      [C] -> 2 logits
#include "lrg_common.h"
#include "lrg_fused.h"
#ifndef LRG_FEW_TILES_SLOTS
#define LRG_FEW_TILES_SLOTS 64       // packed launches over at most this many instances count as "few tiles" (LrgFusedArgs.few)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LRG_BM 64
#define LRG_BN 64
#define LRG_BK 64
#define LRG_LDA (LRG_BK + 4)   // +4 floats: conflict-free ds_read_b128 of A fragments, rows stay 16-B aligned
#define LRG_LDB (LRG_BN)

struct LrgLayerProb {
    const float *x;
    const float *w;
    const float *bias;
    float *y;
    float *pool;
    long rows;
    int rows_per_inst;
    int pad;
};

struct LrgLayerArgs {
    LrgLayerProb p[2];
    int ldx, ldw, K, N;
    int relu, bias_inst_stride, pool_stride, ncol;
    int vec_x, vec_w;
};

// One 64x64 output tile per 256-thread workgroup (4 waves as 2x2 of 32x32), K streamed through LDS in
// chunks of 64.  Row tiles are dealt to XCDs (block b runs on XCD b%8) so that all column tiles of one row
// tile hit the same per-XCD L2.
__global__ __launch_bounds__(256) void lrg_pointwise_mfma_kernel(LrgLayerArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[LRG_BM * LRG_LDA + LRG_BK * LRG_LDB];
    float *As = smem;
    float *Bs = smem + LRG_BM * LRG_LDA;

    const LrgLayerProb P = a.p[blockIdx.y];
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    const int rt = (q / a.ncol) * 8 + xcd;
    const int ct = q % a.ncol;
    const long r0 = (long)rt * LRG_BM;
    if (r0 >= P.rows) return;
    const int n0 = ct * LRG_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int K = a.K;
    for (int k0 = 0; k0 < K; k0 += LRG_BK) {
        const int kw = min(LRG_BK, K - k0);
        const int kwp = (kw + 7) & ~7;
        // ---- stage A (rows of x) ----
        if (a.vec_x) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                int idx = tid + it * 256;
                int row = idx >> 4, c4 = idx & 15;
                int kk = k0 + 4 * c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r0 + row < P.rows && kk < K) v = *reinterpret_cast<const float4 *>(P.x + (r0 + row) * a.ldx + kk);
                *reinterpret_cast<float4 *>(&As[row * LRG_LDA + 4 * c4]) = v;
            }
        } else {
            for (int idx = tid; idx < LRG_BM * kwp; idx += 256) {
                int row = idx / kwp, c = idx - row * kwp;
                float v = 0.f;
                if (r0 + row < P.rows && c < kw) v = P.x[(r0 + row) * a.ldx + k0 + c];
                As[row * LRG_LDA + c] = v;
            }
        }
        // ---- stage B (rows of W) ----
        if (a.vec_w) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                int idx = tid + it * 256;
                int kr = idx >> 4, c4 = idx & 15;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kr < kw) v = *reinterpret_cast<const float4 *>(P.w + (long)(k0 + kr) * a.ldw + n0 + 4 * c4);
                *reinterpret_cast<float4 *>(&Bs[kr * LRG_LDB + 4 * c4]) = v;
            }
        } else {
            for (int idx = tid; idx < kwp * LRG_BN; idx += 256) {
                int kr = idx >> 6, c = idx & 63;
                float v = 0.f;
                if (kr < kw) v = P.w[(long)(k0 + kr) * a.ldw + n0 + c];
                Bs[kr * LRG_LDB + c] = v;
            }
        }
        __syncthreads();
        // ---- 32x32x2 fp32 MFMA: lane half h feeds logical k = 8g + 4h + s of both operands ----
        const float *ap = &As[(wm * 32 + li) * LRG_LDA + 4 * lh];
        const float *bp = &Bs[(4 * lh) * LRG_LDB + wn * 32 + li];
        const int ng = kwp >> 3;
        for (int g = 0; g < ng; ++g) {
            float4 av = *reinterpret_cast<const float4 *>(ap + 8 * g);
            float b0 = bp[(8 * g + 0) * LRG_LDB];
            float b1 = bp[(8 * g + 1) * LRG_LDB];
            float b2 = bp[(8 * g + 2) * LRG_LDB];
            float b3 = bp[(8 * g + 3) * LRG_LDB];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc, 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
    const int col = n0 + wn * 32 + li;
    const long rbase = r0 + wm * 32 + 4 * lh;
    const bool inst_bias = a.bias_inst_stride > 0 && P.rows_per_inst > 0;
    float bias0 = 0.f;
    if (P.bias && !inst_bias) bias0 = P.bias[col];
    float cmax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        long row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < P.rows) {
            float bv = bias0;
            if (inst_bias) bv = P.bias[(row / P.rows_per_inst) * a.bias_inst_stride + col];
            float v = acc[r] + bv;
            if (a.relu) v = fmaxf(v, 0.f);
            P.y[row * a.N + col] = v;
            cmax = fmaxf(cmax, v);
        }
    }
    if (P.pool) {   // host guarantees relu and rows_per_inst % 64 == 0: the whole tile is one instance
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        if (lh == 0) atomicMax(reinterpret_cast<int *>(P.pool + (r0 / P.rows_per_inst) * a.pool_stride + col),
                               __float_as_int(cmax));
    }
}

// Any-shape fallback (cout not a multiple of 64): one thread per output element.
__global__ void lrg_pointwise_naive_kernel(LrgLayerArgs a) {
    const LrgLayerProb P = a.p[blockIdx.y];
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.rows * a.N) return;
    long row = e / a.N;
    int col = (int)(e - row * a.N);
    float s = 0.f;
    for (int k = 0; k < a.K; ++k) s = fmaf(P.x[row * a.ldx + k], P.w[(long)k * a.ldw + col], s);
    float bv = 0.f;
    if (P.bias) bv = (a.bias_inst_stride > 0 && P.rows_per_inst > 0)
                         ? P.bias[(row / P.rows_per_inst) * a.bias_inst_stride + col] : P.bias[col];
    s += bv;
    if (a.relu) s = fmaxf(s, 0.f);
    P.y[row * a.N + col] = s;
    if (P.pool) atomicMax(reinterpret_cast<int *>(P.pool + (row / P.rows_per_inst) * a.pool_stride + col), __float_as_int(s));
}

static int launch_layer(const LrgLayerArgs &a0, int nprob, hipStream_t st) {
    LrgLayerArgs a = a0;
    long maxrows = a.p[0].rows;
    if (nprob > 1 && a.p[1].rows > maxrows) maxrows = a.p[1].rows;
    if (maxrows <= 0) return 0;
    if (a.N % LRG_BN == 0) {
        a.ncol = a.N / LRG_BN;
        long nrt = (maxrows + LRG_BM - 1) / LRG_BM;
        long nrt8 = (nrt + 7) / 8 * 8;
        dim3 grid((unsigned)(nrt8 * a.ncol), nprob, 1);
        hipLaunchKernelGGL(lrg_pointwise_mfma_kernel, grid, dim3(256), 0, st, a);
    } else {
        long tot = maxrows * a.N;
        dim3 grid((unsigned)((tot + 255) / 256), nprob, 1);
        hipLaunchKernelGGL(lrg_pointwise_naive_kernel, grid, dim3(256), 0, st, a);
    }
    LRG_LAUNCH_CHECK();
    return 0;
}

static void fill_vec_flags(LrgLayerArgs &a, int nprob) {
    a.vec_x = (a.ldx % 4 == 0) && (a.K % 4 == 0);
    a.vec_w = (a.ldw % 4 == 0);
    for (int i = 0; i < nprob; ++i) {
        if (((uintptr_t)a.p[i].x & 15) != 0) a.vec_x = 0;
        if (((uintptr_t)a.p[i].w & 15) != 0) a.vec_w = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// column max over an instance's rows
// ------------------------------------------------------------------------------------------------
struct LrgSegmaxArgs {
    const float *x[2];
    float *out[2];
    int rows[2];
    int C, out_stride;
};

__global__ __launch_bounds__(256) void lrg_segmax_kernel(LrgSegmaxArgs a) {
    __shared__ float red[16][65];
    const int z = blockIdx.z;
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 columns x 4 row groups, 256-B rows per wave
    const int rows = a.rows[z];
    const float *x = a.x[z] + (long)b * rows * a.C;
    float m = -INFINITY;
    if (c0 + tx < a.C)
        for (int r = ty; r < rows; r += 4) m = fmaxf(m, x[(long)r * a.C + c0 + tx]);
    red[ty][tx] = m;
    __syncthreads();
    if (ty == 0 && c0 + tx < a.C) {
        m = fmaxf(fmaxf(red[0][tx], red[1][tx]), fmaxf(red[2][tx], red[3][tx]));
        a.out[z][(long)b * a.out_stride + c0 + tx] = m;
    }
}

// ------------------------------------------------------------------------------------------------
// hoisted pooled-feature product: hb[b,c] = bias[c] + sum_k pooled[b,k] w[k,c]
// ------------------------------------------------------------------------------------------------
#define LRG_GEMV_TB 2

// 64 output columns x TB instances per 512-thread block; the 8 waves split K and their partial sums are
// combined through LDS in a fixed order (deterministic, no atomics).
#define LRG_GEMV_WAVES 8
__global__ __launch_bounds__(64 * LRG_GEMV_WAVES) void lrg_head_gemv_kernel(LrgGemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pl[];   // [TB][P] pooled rows, then [4][TB][64] partials
    const int z = blockIdx.z;
    const int b0 = blockIdx.y * LRG_GEMV_TB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    if (a.cnt_src && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 2) {
        a.cnt_dst[threadIdx.x] = a.cnt_src[threadIdx.x];
        a.cnt_src[threadIdx.x] = 0;
    }
    const int nb = min(LRG_GEMV_TB, a.B - b0);
    for (int i = threadIdx.x; i < LRG_GEMV_TB * a.P; i += 64 * LRG_GEMV_WAVES) {
        int bi = i / a.P;
        pl[i] = bi < nb ? a.pooled[(long)(b0 + bi) * a.P + (i - bi * a.P)] : 0.f;
    }
    __syncthreads();
    float acc[LRG_GEMV_TB];
#pragma unroll
    for (int i = 0; i < LRG_GEMV_TB; ++i) acc[i] = 0.f;
    const int kq = (a.P + LRG_GEMV_WAVES - 1) / LRG_GEMV_WAVES;
    const int k0 = wave * kq, k1 = min(a.P, k0 + kq);
    if (c < a.C) {
        const float *w = a.w[z] + c;
        if (((k1 - k0) & 7) == 0) {
            // k in the order the matrix-core formulation below adds its products: v_mfma_f32_32x32x2_f32 is two chained FMAs, lane
            // half 0 then lane half 1, and the halves of k-group g feed k = 8g + s and 8g + 4 + s (s = 0 .. 3) -- so 8g + 0, 4, 1, 5, 2,
            // 6, 3, 7.  With that, this kernel, lrg_head_gemm_kernel and the free-running kernel's pooled blocks (lrg_async.inl) give
            // the same bits whatever the batch (tools/mfma_order_check.py, profiles/r03_mfma_order.txt: 100 % against 26 % k after k).
            for (int kg = k0; kg < k1; kg += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = kg + (((u & 1) << 2) | (u >> 1));
                    float wv = w[(long)k * a.ldw];
#pragma unroll
                    for (int i = 0; i < LRG_GEMV_TB; ++i) acc[i] = fmaf(pl[i * a.P + k], wv, acc[i]);
                }
            }
        } else {
#pragma unroll 16
            for (int k = k0; k < k1; ++k) {
                float wv = w[(long)k * a.ldw];
#pragma unroll
                for (int i = 0; i < LRG_GEMV_TB; ++i) acc[i] = fmaf(pl[i * a.P + k], wv, acc[i]);
            }
        }
    }
    float *part = pl + LRG_GEMV_TB * a.P;
#pragma unroll
    for (int i = 0; i < LRG_GEMV_TB; ++i) part[(wave * LRG_GEMV_TB + i) * 64 + lane] = acc[i];
    __syncthreads();
    if (wave == 0 && c < a.C) {
        float bv = a.bias[z] ? a.bias[z][c] : 0.f;
        for (int i = 0; i < nb; ++i) {
            float s = part[(0 * LRG_GEMV_TB + i) * 64 + lane];
#pragma unroll
            for (int wv = 1; wv < LRG_GEMV_WAVES; ++wv) s += part[(wv * LRG_GEMV_TB + i) * 64 + lane];
            a.hb[z][(long)(b0 + i) * a.C + c] = s + bv;
        }
    }
}

// The same product as a GEMM on the matrix cores, for batches: a 32-instance x 32-column tile per workgroup, the 8 waves
// split K (each P/8 deep) and their partial tiles are summed through LDS in a fixed order.  The weights are then read once
// per 32 instances instead of once per LRG_GEMV_TB (70 MB -> 6 MB of L2 traffic at 68 instances).
typedef float gemm_f32x16 __attribute__((ext_vector_type(16)));
#ifndef LRG_GEMM_EXCLUSIVE
#define LRG_GEMM_EXCLUSIVE 0
#endif
#define LRG_GEMM_WAVES 8
template <int NG>      // k-groups of 8 per wave: P = 64 * NG; fully unrolled so that every operand load is in flight at once
__global__ __launch_bounds__(64 * LRG_GEMM_WAVES) void lrg_head_gemm_kernel(LrgGemvArgs a) {
    __shared__ float part[LRG_GEMM_WAVES][32][33];
#if LRG_GEMM_EXCLUSIVE
    asm volatile("" ::: "v255");          // accounted 256 VGPRs: the two waves per SIMD of a workgroup fill the register file, a CU to itself
#endif
    const int z = blockIdx.z, c0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    if (a.cnt_src && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 2) {
        a.cnt_dst[threadIdx.x] = a.cnt_src[threadIdx.x];
        a.cnt_src[threadIdx.x] = 0;
    }
    const int k0 = wave * 8 * NG;
    const int row = min(b0 + li, a.B - 1);                                   // rows past the batch repeat the last one
    const float *ap = a.pooled + (long)row * a.P + k0 + 4 * lh;
    const float *wp = a.w[z] + (long)(k0 + 4 * lh) * a.ldw + c0 + li;
    gemm_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // every operand of this wave's K range is requested before the first MFMA (one memory round trip, not NG)
    float4 av[NG];
    float wv[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {                                           // lane half h feeds k = 8g + 4h + s
        av[g] = *reinterpret_cast<const float4 *>(ap + 8 * g);
        const float *w0 = wp + (long)(8 * g) * a.ldw;
        wv[g][0] = w0[0]; wv[g][1] = w0[a.ldw]; wv[g][2] = w0[2 * a.ldw]; wv[g][3] = w0[3 * a.ldw];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].x, wv[g][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].y, wv[g][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].z, wv[g][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].w, wv[g][3], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
    __syncthreads();
    for (int o = threadIdx.x; o < 32 * 32; o += 64 * LRG_GEMM_WAVES) {
        const int r = o >> 5, c = o & 31;
        float s = part[0][r][c];
#pragma unroll
        for (int w = 1; w < LRG_GEMM_WAVES; ++w) s += part[w][r][c];
        if (b0 + r < a.B) a.hb[z][(long)(b0 + r) * a.C + c0 + c] = s + (a.bias[z] ? a.bias[z][c0 + c] : 0.f);
    }
}

// picks the formulation: the matrix-core GEMM for aligned shapes and more than a handful of instances
static int launch_head_gemv(const LrgGemvArgs &g, int nheads, hipStream_t st) {
    const int ng = g.P / (8 * LRG_GEMM_WAVES);
    const bool aligned = g.P == ng * 8 * LRG_GEMM_WAVES && (ng == 16 || ng == 8 || ng == 2) && g.C % 32 == 0 &&
                         (((uintptr_t)g.pooled) & 15) == 0;
    if (aligned && g.B >= 8) {
        const dim3 grid(g.C / 32, (g.B + 31) / 32, nheads), block(64 * LRG_GEMM_WAVES);
        if (ng == 16) hipLaunchKernelGGL(lrg_head_gemm_kernel<16>, grid, block, 0, st, g);        // lite 0: P = 1024
        else if (ng == 8) hipLaunchKernelGGL(lrg_head_gemm_kernel<8>, grid, block, 0, st, g);     // lite 2: P = 512
        else hipLaunchKernelGGL(lrg_head_gemm_kernel<2>, grid, block, 0, st, g);                  // lite 1: P = 128
    } else {
        size_t sh = ((size_t)LRG_GEMV_TB * g.P + LRG_GEMV_WAVES * LRG_GEMV_TB * 64) * sizeof(float);
        hipLaunchKernelGGL(lrg_head_gemv_kernel, dim3((g.C + 63) / 64, (g.B + LRG_GEMV_TB - 1) / LRG_GEMV_TB, nheads),
                           dim3(64 * LRG_GEMV_WAVES), sh, st, g);
    }
    LRG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// final head layer: [rows,C] @ [C,2] + b, no ReLU.  16 lanes per row, float4 loads.
// ------------------------------------------------------------------------------------------------
struct LrgFinalArgs {
    const float *h[2];
    const float *w[2];
    const float *bias[2];
    float *out[2];
    long rows[2];
    int C;
};

__global__ __launch_bounds__(256) void lrg_head_final_kernel(LrgFinalArgs a) {
    const int z = blockIdx.y;
    const int sub = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool valid = row < a.rows[z];
    float s0 = 0.f, s1 = 0.f;
    if (valid) {
        const float *h = a.h[z] + row * a.C;
        const float *w = a.w[z];
        for (int k = 4 * sub; k < a.C; k += 64) {
            float4 hv = *reinterpret_cast<const float4 *>(h + k);
            float4 w01 = *reinterpret_cast<const float4 *>(w + 2 * k);       // w[k][0],w[k][1],w[k+1][0],w[k+1][1]
            float4 w23 = *reinterpret_cast<const float4 *>(w + 2 * k + 4);
            s0 = fmaf(hv.x, w01.x, s0); s1 = fmaf(hv.x, w01.y, s1);
            s0 = fmaf(hv.y, w01.z, s0); s1 = fmaf(hv.y, w01.w, s1);
            s0 = fmaf(hv.z, w23.x, s0); s1 = fmaf(hv.z, w23.y, s1);
            s0 = fmaf(hv.w, w23.z, s0); s1 = fmaf(hv.w, w23.w, s1);
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o);
        s1 += __shfl_xor(s1, o);
    }
    if (valid && sub == 0) {
        a.out[z][row * 2 + 0] = s0 + a.bias[z][0];
        a.out[z][row * 2 + 1] = s1 + a.bias[z][1];
    }
}

// ------------------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------------------
struct LrgFwdLayout {
    size_t conv[2][LRG_MAX_CONV];   // float offsets
    size_t pooled;
    size_t hb[2];                   // 0 add, 1 remove
    size_t hid[2][LRG_MAX_HEAD];
    size_t scratch;                 // 64 floats, reserved
    size_t packed;                  // lrg_pack_weights image, used when the caller supplies none
    size_t tiles;                   // live-tile lists of lrg_forward_rows (int32; see lrg_hip.h, view kind 6)
    size_t tiles_cap[2];            // entries of the inlier / neighbour list
    size_t total;
    int P;                          // 2*C_last
};

// ------------------------------------------------------------------------------------------------
// MFMA-operand image of the [Cin,Cout] kernels (lrg_pack_weights).  Per layer: for each 32-column block, for each
// group g of 8 rows (Cin zero-padded to a multiple of 8), 64 lanes x float4:
//   image[((cb * ng + g) * 64 + lane) * 4 + s] = W[8g + 4*(lane>>5) + s][32*cb + (lane&31)]
// i.e. exactly the four B values lane `lane` feeds to the four v_mfma_f32_32x32x2_f32 of k-group g.
// ------------------------------------------------------------------------------------------------
struct LrgPackLayout {
    size_t conv[2][LRG_MAX_CONV];   // float offsets; 0 inlier, 1 neighbour branch
    size_t head[2][LRG_MAX_HEAD];   // 0 add, 1 remove; j = 0 covers only the conv[1] rows of the first kernel
    size_t total;
};

static inline int lrg_kgroups(int K) { return (K + 7) >> 3; }

static int pack_layout(const LrgWeights *w, LrgPackLayout *L) {
    if (!w) return LRG_EINVAL - 1;
    if (w->n_conv < 2 || w->n_conv > LRG_MAX_CONV || w->n_head < 2 || w->n_head > LRG_MAX_HEAD) return LRG_EINVAL - 2;
    size_t off = 0;
    for (int br = 0; br < 2; ++br)
        for (int i = 0; i < w->n_conv; ++i) {
            const int K = i == 0 ? w->feature_size : w->conv_ch[i - 1];
            L->conv[br][i] = off;
            off = lrg_align_up(off + (size_t)lrg_kgroups(K) * 8 * lrg_align_up(w->conv_ch[i], 32), 64);
        }
    for (int hd = 0; hd < 2; ++hd)
        for (int j = 0; j < w->n_head - 1; ++j) {
            const int K = j == 0 ? w->conv_ch[1] : w->head_ch[j - 1];
            L->head[hd][j] = off;
            off = lrg_align_up(off + (size_t)lrg_kgroups(K) * 8 * lrg_align_up(w->head_ch[j], 32), 64);
        }
    L->total = off;
    return 0;
}

#define LRG_PACK_MAX (2 * LRG_MAX_CONV + 2 * LRG_MAX_HEAD)
struct LrgPackArgs {
    const float *src[LRG_PACK_MAX];
    float *dst[LRG_PACK_MAX];
    int K[LRG_PACK_MAX], N[LRG_PACK_MAX];
};

__global__ __launch_bounds__(256) void lrg_pack_weights_kernel(LrgPackArgs a) {
    const int e = blockIdx.y;
    const float *src = a.src[e];
    const int K = a.K[e], N = a.N[e];
    const int ng = (K + 7) >> 3, ncb = (N + 31) >> 5;
    const long total = (long)ncb * ng * 64;              // float4 slots
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long cg = idx >> 6;
        const int g = (int)(cg % ng), cb = (int)(cg / ng);
        const int col = 32 * cb + (lane & 31), k0 = 8 * g + 4 * (lane >> 5);
        float4 v;
        v.x = (k0 + 0 < K && col < N) ? src[(long)(k0 + 0) * N + col] : 0.f;
        v.y = (k0 + 1 < K && col < N) ? src[(long)(k0 + 1) * N + col] : 0.f;
        v.z = (k0 + 2 < K && col < N) ? src[(long)(k0 + 2) * N + col] : 0.f;
        v.w = (k0 + 3 < K && col < N) ? src[(long)(k0 + 3) * N + col] : 0.f;
        reinterpret_cast<float4 *>(a.dst[e])[idx] = v;
    }
}

static int pack_weights(const LrgWeights *w, float *dst, hipStream_t st) {
    LrgPackLayout PL;
    int rc = pack_layout(w, &PL);
    if (rc) return rc;
    LrgPackArgs a = {};
    int n = 0;
    for (int br = 0; br < 2; ++br)
        for (int i = 0; i < w->n_conv; ++i) {
            a.src[n] = br == 0 ? w->inlier_w[i] : w->neighbor_w[i];
            a.dst[n] = dst + PL.conv[br][i];
            a.K[n] = i == 0 ? w->feature_size : w->conv_ch[i - 1];
            a.N[n] = w->conv_ch[i];
            if (!a.src[n]) return LRG_EINVAL - 40;
            ++n;
        }
    const int P = 2 * w->conv_ch[w->n_conv - 1];
    for (int hd = 0; hd < 2; ++hd)
        for (int j = 0; j < w->n_head - 1; ++j) {
            const float *W = hd == 0 ? w->add_w[j] : w->rmv_w[j];
            if (!W) return LRG_EINVAL - 40;
            a.src[n] = j == 0 ? W + (size_t)P * w->head_ch[0] : W;      // rows P.. of the first kernel: the conv[1] part (:131,:135)
            a.dst[n] = dst + PL.head[hd][j];
            a.K[n] = j == 0 ? w->conv_ch[1] : w->head_ch[j - 1];
            a.N[n] = w->head_ch[j];
            ++n;
        }
    hipLaunchKernelGGL(lrg_pack_weights_kernel, dim3(32, n), dim3(256), 0, st, a);
    LRG_LAUNCH_CHECK();
    return 0;
}

static int fwd_layout(const LrgWeights *w, int B, int ni, int nn, LrgFwdLayout *L) {
    if (!w || B <= 0 || ni <= 0 || nn <= 0) return LRG_EINVAL - 1;
    if (w->n_conv < 2 || w->n_conv > LRG_MAX_CONV || w->n_head < 2 || w->n_head > LRG_MAX_HEAD) return LRG_EINVAL - 2;
    if (w->head_ch[w->n_head - 1] != 2) return LRG_EINVAL - 3;
    size_t off = 0;
    const long rows[2] = {(long)B * ni, (long)B * nn};
    for (int br = 0; br < 2; ++br)
        for (int i = 0; i < w->n_conv; ++i) {
            L->conv[br][i] = off;
            off = lrg_align_up(off + (size_t)rows[br] * w->conv_ch[i], 64);
        }
    L->P = 2 * w->conv_ch[w->n_conv - 1];
    L->pooled = off;
    off = lrg_align_up(off + (size_t)B * L->P, 64);
    for (int hd = 0; hd < 2; ++hd) {
        L->hb[hd] = off;
        off = lrg_align_up(off + (size_t)B * w->head_ch[0], 64);
    }
    // head 0 = add head on the neighbour rows, head 1 = remove head on the inlier rows
    for (int hd = 0; hd < 2; ++hd)
        for (int i = 0; i < w->n_head - 1; ++i) {
            L->hid[hd][i] = off;
            off = lrg_align_up(off + (size_t)rows[hd == 0 ? 1 : 0] * w->head_ch[i], 64);
        }
    L->scratch = off;
    off = lrg_align_up(off + 64, 64);
    LrgPackLayout PL;
    int rc = pack_layout(w, &PL);
    if (rc) return rc;
    L->packed = off;
    off = lrg_align_up(off + PL.total, 64);
    L->tiles = off;
    L->tiles_cap[0] = (size_t)B * ((ni + LRG_ROW_TILE - 1) / LRG_ROW_TILE);
    L->tiles_cap[1] = (size_t)B * ((nn + LRG_ROW_TILE - 1) / LRG_ROW_TILE);
    off = lrg_align_up(off + 2 + L->tiles_cap[0] + L->tiles_cap[1], 64);
    L->total = off;
    return 0;
}

// Fused evaluation: 3 launches (both branches | pooled-feature GEMV | both heads).
static int forward_fused(const LrgWeights *w, const float *inlier, const float *neighbor, int B, int n_inlier,
                         int n_neighbor, const int32_t *rows_in, const int32_t *rows_nb, float *add_logits,
                         float *rmv_logits, float *ws, const LrgFwdLayout &L, bool keep_acts, bool pool_zeroed,
                         bool tile_lists, hipStream_t st) {
    const long rows[2] = {(long)B * n_inlier, (long)B * n_neighbor};
    const int rpi[2] = {n_inlier, n_neighbor};
    const int nc = w->n_conv, nh = w->n_head;
    const int Clast = w->conv_ch[nc - 1];
    if (!pool_zeroed) LRG_HIP_CHECK(hipMemsetAsync(ws + L.pooled, 0, (size_t)B * L.P * sizeof(float), st));
    const int *tl = reinterpret_cast<const int *>(ws + L.tiles);
    LrgPackLayout PL;
    int prc = pack_layout(w, &PL);
    if (prc) return prc;
    const float *pk = static_cast<const float *>(w->packed);
    if (!pk) {                                   // no image from the caller: build one next to the activations
        if ((prc = pack_weights(w, ws + L.packed, st))) return prc;
        pk = ws + L.packed;
    }
    {
        LrgFusedArgs a = {};
        for (int br = 0; br < 2; ++br) {
            LrgFusedProb &P = a.p[br];
            P.x = br == 0 ? inlier : neighbor;
            P.ldx = w->feature_size; P.Kin = w->feature_size;
            P.rows = rows[br]; P.rows_per_inst = rpi[br];
            P.pool = ws + L.pooled + (br == 0 ? 0 : Clast); P.pool_stride = L.P;
            P.valid = br == 0 ? rows_in : rows_nb;
            if (tile_lists) { P.tile_count = tl + br; P.tile_list = tl + 2 + (br == 0 ? 0 : L.tiles_cap[0]); }
            P.nlayers = nc;
            for (int i = 0; i < nc; ++i) {
                LrgFusedLayer &F = P.L[i];
                F.w = pk + PL.conv[br][i];
                F.bias = br == 0 ? w->inlier_b[i] : w->neighbor_b[i];
                F.K = i == 0 ? w->feature_size : w->conv_ch[i - 1];
                F.N = w->conv_ch[i]; F.ng = lrg_kgroups(F.K);
                F.flags = LRG_FL_RELU | (i + 1 < nc ? LRG_FL_KEEP : LRG_FL_POOL);
                F.gout = (i == 1 || keep_acts) ? ws + L.conv[br][i] : nullptr;      // conv[1] feeds the heads (:130,:134)
            }
        }
        int rc = lrg_fused_branches(a, 2, st);
        if (rc) return rc;
    }
    const int C0 = w->head_ch[0];
    {
        LrgGemvArgs g = {};
        g.pooled = ws + L.pooled;
        g.w[0] = w->add_w[0]; g.w[1] = w->rmv_w[0];
        g.bias[0] = w->add_b[0]; g.bias[1] = w->rmv_b[0];
        g.hb[0] = ws + L.hb[0]; g.hb[1] = ws + L.hb[1];
        g.ldw = C0; g.B = B; g.P = L.P; g.C = C0;
        int grc = launch_head_gemv(g, 2, st);
        if (grc) return grc;
    }
    {
        const int hbr[2] = {1, 0};      // head 0 = add on the neighbour rows, head 1 = remove on the inlier rows
        LrgFusedArgs a = {};
        for (int hd = 0; hd < 2; ++hd) {
            const int br = hbr[hd];
            LrgFusedProb &P = a.p[hd];
            P.x = ws + L.conv[br][1];
            P.ldx = w->conv_ch[1]; P.Kin = w->conv_ch[1];
            P.rows = rows[br]; P.rows_per_inst = rpi[br];
            P.valid = br == 0 ? rows_in : rows_nb;
            if (tile_lists) { P.tile_count = tl + br; P.tile_list = tl + 2 + (br == 0 ? 0 : L.tiles_cap[0]); }
            P.nlayers = nh - 1;
            for (int i = 0; i < nh - 1; ++i) {
                LrgFusedLayer &F = P.L[i];
                F.N = w->head_ch[i];
                F.w = pk + PL.head[hd][i];           // i = 0: rows 2*C_last.. of W0, the conv[1] part of the concat (:131,:135)
                if (i == 0) {
                    F.bias = ws + L.hb[hd];
                    F.K = w->conv_ch[1];
                    F.flags = LRG_FL_RELU | LRG_FL_KEEP | LRG_FL_INST_BIAS;
                } else {
                    F.bias = hd == 0 ? w->add_b[i] : w->rmv_b[i];
                    F.K = w->head_ch[i - 1];
                    F.flags = LRG_FL_RELU | LRG_FL_KEEP;
                    // the head kernel's odd buffer only holds the 64-wide input: a wider last hidden layer is
                    // written over its own (dead) input instead
                    if (i == nh - 2 && (i & 1) && F.N > 64) F.flags |= LRG_FL_INPLACE;
                }
                F.ng = lrg_kgroups(F.K);
                F.gout = keep_acts ? ws + L.hid[hd][i] : nullptr;
            }
            P.fw = hd == 0 ? w->add_w[nh - 1] : w->rmv_w[nh - 1];
            P.fb = hd == 0 ? w->add_b[nh - 1] : w->rmv_b[nh - 1];
            P.fout = hd == 0 ? add_logits : rmv_logits;
            if (pool_zeroed && hd == 0) { P.zero_pool = ws + L.pooled; P.zero_count = L.P; }
        }
        return lrg_fused_heads(a, 2, st);
    }
}

// ------------------------------------------------------------------------------------------------
// packed-row evaluation (the grow loop's formulation): the distinct rows of all instances back to back
// ------------------------------------------------------------------------------------------------
struct LrgPackedLayout {
    size_t conv1[2];     // float offsets: conv[1] of the packed inlier / neighbour rows (:130,:134)
    size_t pooled;       // [n_inst, 2*C_last]
    size_t hb[2];        // [n_inst, head_ch[0]] hoisted pooled product of the add / remove head
    size_t packed;       // lrg_pack_weights image when the caller supplies none
    size_t conv3[2];     // the last layer before the pooled one, [row_cap, conv_ch[n_conv - 2]] per side: from the PREFIX to the POOL tasks of a wave-branch launch (lrg_wave_tile.inl)
    size_t total;
    int P;
};

static int packed_layout(const LrgWeights *w, int n_inst, int row_cap, LrgPackedLayout *L) {
    if (!w || n_inst <= 0 || row_cap <= 0 || row_cap % LRG_ROW_TILE != 0) return LRG_EINVAL - 1;
    if (w->n_conv < 2 || w->n_conv > LRG_MAX_CONV || w->n_head < 2 || w->n_head > LRG_MAX_HEAD) return LRG_EINVAL - 2;
    if (w->head_ch[w->n_head - 1] != 2) return LRG_EINVAL - 3;
    size_t off = 0;
    for (int br = 0; br < 2; ++br) { L->conv1[br] = off; off = lrg_align_up(off + (size_t)row_cap * w->conv_ch[1], 64); }
    L->P = 2 * w->conv_ch[w->n_conv - 1];
    L->pooled = off;
    off = lrg_align_up(off + (size_t)n_inst * L->P, 64);
    for (int hd = 0; hd < 2; ++hd) { L->hb[hd] = off; off = lrg_align_up(off + (size_t)n_inst * w->head_ch[0], 64); }
    LrgPackLayout PL;
    int rc = pack_layout(w, &PL);
    if (rc) return rc;
    L->packed = off;
    off = lrg_align_up(off + PL.total, 64);
    for (int br = 0; br < 2; ++br) { L->conv3[br] = off; off = lrg_align_up(off + (size_t)row_cap * w->conv_ch[w->n_conv - 2], 64); }
    L->total = off;
    return 0;
}

int lrg_packed_conv3_view(const LrgWeights *w, int n_inst, int row_cap, size_t offset_floats[2]) {
    LrgPackedLayout L;
    int rc = packed_layout(w, n_inst, row_cap, &L);
    if (rc) return rc;
    offset_floats[0] = L.conv3[0]; offset_floats[1] = L.conv3[1];
    return 0;
}

// the shapes the fused kernels are instantiated for (lite 0/1/2), as lrg_forward_rows checks them
static int packed_shapes(const LrgWeights *w) {
    const int nc = w->n_conv, nh = w->n_head;
    for (int i = 0; i < nc; ++i)
        if (w->conv_ch[i] % 64 != 0 || (i + 1 < nc && w->conv_ch[i] > 128)) return LRG_EINVAL - 7;
    for (int i = 0; i < nh - 1; ++i)
        if (w->head_ch[i] % 64 != 0 || w->head_ch[i] > ((i & 1) ? 128 : 256) || ((i & 1) && w->head_ch[i] > 64 && i != nh - 2))
            return LRG_EINVAL - 7;
    return 0;
}

// The problems of a packed evaluation: the two branch stacks, the pooled GEMM and the two head stacks, as the launchers (and the
// free-running kernel, lrg_async.inl) take them.  `pk` = the packed weight image in use.
static int packed_problems(const LrgWeights *w, const float *pk, const LrgPackLayout &PL, const float *x_in, const float *x_nb, const float *center,
                           const int32_t *row_inst_in, const int32_t *row_inst_nb, int32_t *nrows, int32_t *nrows_heads, int n_inst, int row_cap,
                           float *add_logits, float *rmv_logits, float *ws, const LrgPackedLayout &L, LrgFusedArgs *branches, LrgGemvArgs *gemv,
                           LrgFusedArgs *heads) {
    const int nc = w->n_conv, nh = w->n_head;
    const int Clast = w->conv_ch[nc - 1];
    {
        LrgFusedArgs a = {};
        a.few = n_inst <= LRG_FEW_TILES_SLOTS ? 1 : 0;      // few tiles per launch: two workgroups per CU at most (lrg_fused.hip)
        for (int br = 0; br < 2; ++br) {
            LrgFusedProb &P = a.p[br];
            P.x = br == 0 ? x_in : x_nb;
            P.ldx = w->feature_size; P.Kin = w->feature_size;
            P.rows = row_cap; P.rows_per_inst = row_cap;
            P.nrows = nrows + br; P.row_inst = br == 0 ? row_inst_in : row_inst_nb;
            P.center = center;
            P.pool = ws + L.pooled + (br == 0 ? 0 : Clast); P.pool_stride = L.P;
            P.nlayers = nc;
            for (int i = 0; i < nc; ++i) {
                LrgFusedLayer &F = P.L[i];
                F.w = pk + PL.conv[br][i];
                F.bias = br == 0 ? w->inlier_b[i] : w->neighbor_b[i];
                F.K = i == 0 ? w->feature_size : w->conv_ch[i - 1];
                F.N = w->conv_ch[i]; F.ng = lrg_kgroups(F.K);
                F.flags = LRG_FL_RELU | (i + 1 < nc ? LRG_FL_KEEP : LRG_FL_POOL);
                F.gout = i == 1 ? ws + L.conv1[br] : nullptr;
            }
        }
        *branches = a;
    }
    const int C0 = w->head_ch[0];
    {
        LrgGemvArgs g = {};
        g.pooled = ws + L.pooled;
        g.w[0] = w->add_w[0]; g.w[1] = w->rmv_w[0];
        g.bias[0] = w->add_b[0]; g.bias[1] = w->rmv_b[0];
        g.hb[0] = ws + L.hb[0]; g.hb[1] = ws + L.hb[1];
        g.ldw = C0; g.B = n_inst; g.P = L.P; g.C = C0;
        if (nrows_heads) { g.cnt_src = nrows; g.cnt_dst = nrows_heads; }
        *gemv = g;
    }
    {
        const int hbr[2] = {1, 0};      // head 0 = add on the neighbour rows, head 1 = remove on the inlier rows
        const int32_t *cnt = nrows_heads ? nrows_heads : nrows;
        LrgFusedArgs a = {};
        a.few = n_inst <= LRG_FEW_TILES_SLOTS ? 1 : 0;      // few tiles per launch: two workgroups per CU at most (lrg_fused.hip)
        for (int hd = 0; hd < 2; ++hd) {
            const int br = hbr[hd];
            LrgFusedProb &P = a.p[hd];
            P.x = ws + L.conv1[br];
            P.ldx = w->conv_ch[1]; P.Kin = w->conv_ch[1];
            P.rows = row_cap; P.rows_per_inst = row_cap;
            P.nrows = cnt + br; P.row_inst = br == 0 ? row_inst_in : row_inst_nb;
            P.nlayers = nh - 1;
            for (int i = 0; i < nh - 1; ++i) {
                LrgFusedLayer &F = P.L[i];
                F.N = w->head_ch[i];
                F.w = pk + PL.head[hd][i];
                if (i == 0) {
                    F.bias = ws + L.hb[hd];
                    F.K = w->conv_ch[1];
                    F.flags = LRG_FL_RELU | LRG_FL_KEEP | LRG_FL_INST_BIAS;
                } else {
                    F.bias = hd == 0 ? w->add_b[i] : w->rmv_b[i];
                    F.K = w->head_ch[i - 1];
                    F.flags = LRG_FL_RELU | LRG_FL_KEEP;
                    if (i == nh - 2 && (i & 1) && F.N > 64) F.flags |= LRG_FL_INPLACE;
                }
                F.ng = lrg_kgroups(F.K);
            }
            P.fw = hd == 0 ? w->add_w[nh - 1] : w->rmv_w[nh - 1];
            P.fb = hd == 0 ? w->add_b[nh - 1] : w->rmv_b[nh - 1];
            P.fout = hd == 0 ? add_logits : rmv_logits;
        }
        *heads = a;
    }
    return 0;
}

static int forward_packed(const LrgWeights *w, const float *x_in, const float *x_nb, const float *center, const int32_t *row_inst_in,
                          const int32_t *row_inst_nb, int32_t *nrows, int32_t *nrows_heads, int n_inst, int row_cap,
                          float *add_logits, float *rmv_logits, float *ws, const LrgPackedLayout &L, bool pool_zeroed,
                          hipStream_t st) {
    if (!pool_zeroed) LRG_HIP_CHECK(hipMemsetAsync(ws + L.pooled, 0, (size_t)n_inst * L.P * sizeof(float), st));
    LrgPackLayout PL;
    int prc = pack_layout(w, &PL);
    if (prc) return prc;
    const float *pk = static_cast<const float *>(w->packed);
    if (!pk) {
        if ((prc = pack_weights(w, ws + L.packed, st))) return prc;
        pk = ws + L.packed;
    }
    LrgFusedArgs branches, heads;
    LrgGemvArgs gemv;
    packed_problems(w, pk, PL, x_in, x_nb, center, row_inst_in, row_inst_nb, nrows, nrows_heads, n_inst, row_cap, add_logits, rmv_logits, ws, L,
                    &branches, &gemv, &heads);
    int rc = lrg_fused_branches_packed(branches, 2, st);
    if (rc) return rc;
    if ((rc = launch_head_gemv(gemv, 2, st))) return rc;
    return lrg_fused_heads_packed(heads, 2, st);
}

// (lrg_fused.h) the same descriptors for a caller that runs the tiles itself; needs w->packed
int lrg_packed_problems(const LrgWeights *w, const float *x_in, const float *x_nb, const float *center, const int32_t *row_inst_in,
                        const int32_t *row_inst_nb, int32_t *nrows, int n_inst, int row_cap, float *add_logits, float *rmv_logits,
                        void *workspace, size_t workspace_bytes, LrgFusedArgs *branches, LrgGemvArgs *gemv, LrgFusedArgs *heads) {
    LrgPackedLayout L;
    int rc = packed_layout(w, n_inst, row_cap, &L);
    if (rc) return rc;
    if (!w->packed || !workspace || workspace_bytes < L.total * sizeof(float) || ((uintptr_t)workspace & 255)) return LRG_EINVAL - 6;
    if ((rc = packed_shapes(w))) return rc;
    LrgPackLayout PL;
    if ((rc = pack_layout(w, &PL))) return rc;
    return packed_problems(w, static_cast<const float *>(w->packed), PL, x_in, x_nb, center, row_inst_in, row_inst_nb, nrows, nullptr, n_inst,
                           row_cap, add_logits, rmv_logits, static_cast<float *>(workspace), L, branches, gemv, heads);
}

extern "C" {

size_t lrg_forward_packed_workspace_bytes(const LrgWeights *w, int n_inst, int row_cap) {
    LrgPackedLayout L;
    if (packed_layout(w, n_inst, row_cap, &L) != 0) return 0;
    return L.total * sizeof(float);
}

int lrg_forward_packed_pooled_view(const LrgWeights *w, int n_inst, int row_cap, size_t *offset_floats, size_t *count_floats) {
    LrgPackedLayout L;
    int rc = packed_layout(w, n_inst, row_cap, &L);
    if (rc) return rc;
    if (!offset_floats || !count_floats) return LRG_EINVAL - 4;
    *offset_floats = L.pooled;
    *count_floats = (size_t)n_inst * L.P;
    return 0;
}

int lrg_forward_packed(const LrgWeights *w, const float *x_in, const float *x_nb, const float *center, const int32_t *row_inst_in,
                       const int32_t *row_inst_nb, int32_t *nrows, int32_t *nrows_heads, int n_inst, int row_cap,
                       float *add_logits, float *rmv_logits, void *workspace, size_t workspace_bytes, unsigned flags,
                       void *stream) {
    LrgPackedLayout L;
    int rc = packed_layout(w, n_inst, row_cap, &L);
    if (rc) return rc;
    if (!x_in || !x_nb || !row_inst_in || !row_inst_nb || !nrows || !add_logits || !rmv_logits || !workspace) return LRG_EINVAL - 5;
    if (workspace_bytes < L.total * sizeof(float) || ((uintptr_t)workspace & 255)) return LRG_EINVAL - 6;
    if ((rc = packed_shapes(w))) return rc;
    return forward_packed(w, x_in, x_nb, center, row_inst_in, row_inst_nb, nrows, nrows_heads, n_inst, row_cap, add_logits, rmv_logits,
                          static_cast<float *>(workspace), L, (flags & LRG_FWD_POOL_ZEROED) != 0, (hipStream_t)stream);
}

}  // extern "C"

extern "C" {

int lrg_abi_version(void) { return LRG_ABI_VERSION; }
const char *lrg_target_arch(void) { return "gfx950"; }
size_t lrg_struct_size(int which) {
    switch (which) {
    case 0: return sizeof(LrgWeights);
    case 1: return sizeof(LrgRoom);
    case 2: return sizeof(LrgSlot);
    case 3: return sizeof(LrgGrowParams);
    case 4: return sizeof(LrgStepBuffers);
    case 5: return sizeof(LrgPackedBuffers);
    case 6: return sizeof(LrgBeamGroup);
    case 7: return sizeof(LrgAsyncBuffers);
    case 8: return sizeof(LrgFillJob);
    }
    return 0;
}

size_t lrg_packed_weights_bytes(const LrgWeights *w) {
    LrgPackLayout PL;
    if (pack_layout(w, &PL) != 0) return 0;
    return PL.total * sizeof(float);
}

int lrg_pack_weights(const LrgWeights *w, void *packed, size_t packed_bytes, void *stream) {
    LrgPackLayout PL;
    int rc = pack_layout(w, &PL);
    if (rc) return rc;
    if (!packed || packed_bytes < PL.total * sizeof(float)) return LRG_EINVAL - 41;
    if (((uintptr_t)packed & 255) != 0) return LRG_EINVAL - 42;
    return pack_weights(w, static_cast<float *>(packed), (hipStream_t)stream);
}

size_t lrg_forward_workspace_bytes(const LrgWeights *w, int B, int n_inlier, int n_neighbor) {
    LrgFwdLayout L;
    if (fwd_layout(w, B, n_inlier, n_neighbor, &L) != 0) return 0;
    return L.total * sizeof(float);
}

int lrg_forward_workspace_view(const LrgWeights *w, int B, int n_inlier, int n_neighbor, int kind, int index,
                               size_t *offset_floats, size_t *count_floats) {
    LrgFwdLayout L;
    int rc = fwd_layout(w, B, n_inlier, n_neighbor, &L);
    if (rc) return rc;
    if (!offset_floats || !count_floats) return LRG_EINVAL - 4;
    switch (kind) {
    case 0: case 1:
        if (index < 0 || index >= w->n_conv) return LRG_EINVAL - 5;
        *offset_floats = L.conv[kind][index];
        *count_floats = (size_t)B * (kind == 0 ? n_inlier : n_neighbor) * w->conv_ch[index];
        return 0;
    case 2:
        *offset_floats = L.pooled; *count_floats = (size_t)B * L.P; return 0;
    case 6:
        *offset_floats = L.tiles; *count_floats = 2 + L.tiles_cap[0] + L.tiles_cap[1]; return 0;
    case 5:
        *offset_floats = L.scratch; *count_floats = 64; return 0;
    case 3: case 4:
        if (index < 0 || index >= w->n_head - 1) return LRG_EINVAL - 5;
        *offset_floats = L.hid[kind - 3][index];
        *count_floats = (size_t)B * (kind == 3 ? n_neighbor : n_inlier) * w->head_ch[index];
        return 0;
    }
    return LRG_EINVAL - 6;
}

int lrg_pointwise_layer(const float *x, int ldx, const float *w, int ldw, const float *bias, float *y, long rows,
                        int cin, int cout, int relu, int rows_per_instance, int bias_instance_stride,
                        float *pool_out, int pool_stride, void *stream) {
    if (!x || !w || !y || rows < 0 || cin <= 0 || cout <= 0 || ldx < cin || ldw < cout) return LRG_EINVAL - 1;
    if (pool_out && (!relu || rows_per_instance <= 0 || (cout % LRG_BN == 0 && rows_per_instance % LRG_BM != 0)))
        return LRG_EINVAL - 2;
    LrgLayerArgs a = {};
    a.p[0].x = x; a.p[0].w = w; a.p[0].bias = bias; a.p[0].y = y; a.p[0].pool = pool_out;
    a.p[0].rows = rows; a.p[0].rows_per_inst = rows_per_instance;
    a.ldx = ldx; a.ldw = ldw; a.K = cin; a.N = cout; a.relu = relu;
    a.bias_inst_stride = bias_instance_stride; a.pool_stride = pool_stride;
    fill_vec_flags(a, 1);
    return launch_layer(a, 1, (hipStream_t)stream);
}

int lrg_segmax(const float *x, float *out, int B, int rows, int C, int out_stride, void *stream) {
    if (!x || !out || B <= 0 || rows <= 0 || C <= 0) return LRG_EINVAL - 1;
    LrgSegmaxArgs a = {};
    a.x[0] = x; a.out[0] = out; a.rows[0] = rows; a.C = C; a.out_stride = out_stride;
    hipLaunchKernelGGL(lrg_segmax_kernel, dim3((C + 63) / 64, B, 1), dim3(256), 0, (hipStream_t)stream, a);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_head_pool_gemv(const float *pooled, const float *w, int ldw, const float *bias, float *hb, int B, int P,
                       int C, void *stream) {
    if (!pooled || !w || !hb || B <= 0 || P <= 0 || C <= 0 || ldw < C) return LRG_EINVAL - 1;
    LrgGemvArgs a = {};
    a.pooled = pooled; a.w[0] = w; a.bias[0] = bias; a.hb[0] = hb; a.ldw = ldw; a.B = B; a.P = P; a.C = C;
    return launch_head_gemv(a, 1, (hipStream_t)stream);
}

int lrg_head_final(const float *h, const float *w, const float *bias, float *logits, long rows, int C, void *stream) {
    if (!h || !w || !bias || !logits || rows < 0 || C <= 0 || C % 4 != 0) return LRG_EINVAL - 1;
    if (rows == 0) return 0;
    LrgFinalArgs a = {};
    a.h[0] = h; a.w[0] = w; a.bias[0] = bias; a.out[0] = logits; a.rows[0] = rows; a.C = C;
    hipLaunchKernelGGL(lrg_head_final_kernel, dim3((unsigned)((rows + 15) / 16), 1, 1), dim3(256), 0,
                       (hipStream_t)stream, a);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_forward(const LrgWeights *w, const float *inlier, const float *neighbor, int B, int n_inlier,
                int n_neighbor, float *add_logits, float *rmv_logits, void *workspace, size_t workspace_bytes,
                unsigned flags, void *stream) {
    return lrg_forward_rows(w, inlier, neighbor, B, n_inlier, n_neighbor, nullptr, nullptr, add_logits, rmv_logits,
                            workspace, workspace_bytes, flags, stream);
}

int lrg_forward_rows(const LrgWeights *w, const float *inlier, const float *neighbor, int B, int n_inlier,
                     int n_neighbor, const int32_t *rows_in, const int32_t *rows_nb, float *add_logits,
                     float *rmv_logits, void *workspace, size_t workspace_bytes, unsigned flags, void *stream) {
    LrgFwdLayout L;
    int rc = fwd_layout(w, B, n_inlier, n_neighbor, &L);
    if (rc) return rc;
    if (!inlier || !neighbor || !add_logits || !rmv_logits || !workspace) return LRG_EINVAL - 7;
    if (workspace_bytes < L.total * sizeof(float)) return LRG_EINVAL - 8;
    if (((uintptr_t)workspace & 255) != 0) return LRG_EINVAL - 9;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    const long rows[2] = {(long)B * n_inlier, (long)B * n_neighbor};
    const int rpi[2] = {n_inlier, n_neighbor};
    const int nc = w->n_conv, nh = w->n_head;
    const int Clast = w->conv_ch[nc - 1];
    bool fuse_pool = (flags & LRG_FWD_FUSE_POOL) && (Clast % LRG_BN == 0) && (n_inlier % LRG_BM == 0) &&
                     (n_neighbor % LRG_BM == 0);
    bool fused = (flags & LRG_FWD_FUSED) && (n_inlier % 64 == 0) && (n_neighbor % 64 == 0);
    for (int i = 0; fused && i < nc; ++i)
        if (w->conv_ch[i] % 64 != 0 || (i + 1 < nc && w->conv_ch[i] > 128)) fused = false;
    for (int i = 0; fused && i < nh - 1; ++i)
        if (w->head_ch[i] % 64 != 0 || w->head_ch[i] > ((i & 1) ? 128 : 256) ||
            ((i & 1) && w->head_ch[i] > 64 && i != nh - 2)) fused = false;
    if ((rows_in == nullptr) != (rows_nb == nullptr)) return LRG_EINVAL - 11;
    if (rows_in && !fused) return LRG_EINVAL - 12;      // row counts are honoured by the fused kernels only
    if (fused) return forward_fused(w, inlier, neighbor, B, n_inlier, n_neighbor, rows_in, rows_nb, add_logits, rmv_logits,
                                    ws, L, (flags & LRG_FWD_KEEP_ACTS) != 0, (flags & LRG_FWD_POOL_ZEROED) != 0,
                                    (flags & LRG_FWD_TILE_LISTS) != 0 && rows_in != nullptr && !(flags & LRG_FWD_KEEP_ACTS), st);
    if (fuse_pool) LRG_HIP_CHECK(hipMemsetAsync(ws + L.pooled, 0, (size_t)B * L.P * sizeof(float), st));
    // LRG_FWD_STREAM_TILES: the layer launches on the fused stacks' tile (shapes as for the fused path; else the 64 x 64-tile kernel below)
    bool stream_tiles = (flags & LRG_FWD_STREAM_TILES) && !fuse_pool && (n_inlier % 64 == 0) && (n_neighbor % 64 == 0);
    for (int i = 0; stream_tiles && i < nc; ++i)
        if (w->conv_ch[i] % 64 != 0 || w->conv_ch[i] > 512 || (i + 1 < nc && w->conv_ch[i] > 256)) stream_tiles = false;
    for (int i = 0; stream_tiles && i < nh - 1; ++i)
        if (w->head_ch[i] % 64 != 0 || w->head_ch[i] > 512 || (i + 1 < nh - 1 && w->head_ch[i] != 64 && w->head_ch[i] != 128 && w->head_ch[i] != 256)) stream_tiles = false;
    if (stream_tiles && w->conv_ch[1] != 64 && w->conv_ch[1] != 128) stream_tiles = false;
    const float *pk = nullptr;
    LrgPackLayout PL;
    if (stream_tiles) {
        if ((rc = pack_layout(w, &PL))) return rc;
        pk = static_cast<const float *>(w->packed);
        if (!pk) {
            if ((rc = pack_weights(w, ws + L.packed, st))) return rc;
            pk = ws + L.packed;
        }
    }

    // ---- branches (:106-119): both branches in one launch per layer ----
    for (int i = 0; i < nc; ++i) {
        const int cin_t = i == 0 ? w->feature_size : w->conv_ch[i - 1];
        if (stream_tiles) {
            LrgFusedArgs fa = {};
            for (int br = 0; br < 2; ++br) {
                LrgFusedProb &P = fa.p[br];
                P.x = i == 0 ? (br == 0 ? inlier : neighbor) : ws + L.conv[br][i - 1];
                P.ldx = cin_t; P.Kin = cin_t;
                P.rows = rows[br]; P.rows_per_inst = rpi[br];
                P.nlayers = 1;
                LrgFusedLayer &F = P.L[0];
                F.w = pk + PL.conv[br][i];
                F.bias = br == 0 ? w->inlier_b[i] : w->neighbor_b[i];
                F.K = cin_t; F.N = w->conv_ch[i]; F.ng = lrg_kgroups(F.K);
                F.flags = LRG_FL_RELU;
                F.gout = ws + L.conv[br][i];
            }
            rc = lrg_fused_layer(fa, 2, st);
            if (rc) return rc;
            continue;
        }
        LrgLayerArgs a = {};
        const int cin = i == 0 ? w->feature_size : w->conv_ch[i - 1];
        for (int br = 0; br < 2; ++br) {
            a.p[br].x = i == 0 ? (br == 0 ? inlier : neighbor) : ws + L.conv[br][i - 1];
            a.p[br].w = br == 0 ? w->inlier_w[i] : w->neighbor_w[i];
            a.p[br].bias = br == 0 ? w->inlier_b[i] : w->neighbor_b[i];
            a.p[br].y = ws + L.conv[br][i];
            a.p[br].rows = rows[br];
            a.p[br].rows_per_inst = rpi[br];
            a.p[br].pool = (fuse_pool && i == nc - 1) ? ws + L.pooled + (br == 0 ? 0 : Clast) : nullptr;
        }
        a.ldx = cin; a.ldw = w->conv_ch[i]; a.K = cin; a.N = w->conv_ch[i]; a.relu = 1;
        a.pool_stride = L.P;
        fill_vec_flags(a, 2);
        rc = launch_layer(a, 2, st);
        if (rc) return rc;
    }
    // ---- max-pool + concat (:122-125): pooled = [max(inlier) | max(neighbour)] ----
    if (!fuse_pool) {
        LrgSegmaxArgs s = {};
        for (int br = 0; br < 2; ++br) {
            s.x[br] = ws + L.conv[br][nc - 1];
            s.out[br] = ws + L.pooled + (br == 0 ? 0 : Clast);
            s.rows[br] = rpi[br];
        }
        s.C = Clast; s.out_stride = L.P;
        hipLaunchKernelGGL(lrg_segmax_kernel, dim3((Clast + 63) / 64, B, 2), dim3(256), 0, st, s);
        LRG_LAUNCH_CHECK();
    }
    // ---- heads (:128-162).  head 0 = add on neighbour rows, head 1 = remove on inlier rows.
    // x @ W0 = pooled @ W0[:P] (once per instance) + conv[1] @ W0[P:]  -- the tile/concat is never materialised.
    const int C0 = w->head_ch[0];
    const int Cloc = w->conv_ch[1];
    {
        LrgGemvArgs g = {};
        g.pooled = ws + L.pooled;
        g.w[0] = w->add_w[0]; g.w[1] = w->rmv_w[0];
        g.bias[0] = w->add_b[0]; g.bias[1] = w->rmv_b[0];
        g.hb[0] = ws + L.hb[0]; g.hb[1] = ws + L.hb[1];
        g.ldw = C0; g.B = B; g.P = L.P; g.C = C0;
        int grc = launch_head_gemv(g, 2, st);
        if (grc) return grc;
    }
    const int hbr[2] = {1, 0};   // branch feeding each head
    for (int i = 0; i < nh - 1; ++i) {
        if (stream_tiles) {
            LrgFusedArgs fa = {};
            for (int hd = 0; hd < 2; ++hd) {
                const int br = hbr[hd];
                LrgFusedProb &P = fa.p[hd];
                P.x = i == 0 ? ws + L.conv[br][1] : ws + L.hid[hd][i - 1];
                P.Kin = i == 0 ? Cloc : w->head_ch[i - 1]; P.ldx = P.Kin;
                P.rows = rows[br]; P.rows_per_inst = rpi[br];
                P.nlayers = 1;
                LrgFusedLayer &F = P.L[0];
                F.w = pk + PL.head[hd][i];
                F.bias = i == 0 ? ws + L.hb[hd] : (hd == 0 ? w->add_b[i] : w->rmv_b[i]);
                F.K = P.Kin; F.N = w->head_ch[i]; F.ng = lrg_kgroups(F.K);
                F.flags = LRG_FL_RELU | (i == 0 ? LRG_FL_INST_BIAS : 0);
                F.gout = ws + L.hid[hd][i];
            }
            rc = lrg_fused_layer(fa, 2, st);
            if (rc) return rc;
            continue;
        }
        LrgLayerArgs a = {};
        for (int hd = 0; hd < 2; ++hd) {
            const int br = hbr[hd];
            const float *W = hd == 0 ? w->add_w[i] : w->rmv_w[i];
            if (i == 0) {
                a.p[hd].x = ws + L.conv[br][1];
                a.p[hd].w = W + (size_t)L.P * C0;          // rows P.. of W0: the conv[1] part of the concat
                a.p[hd].bias = ws + L.hb[hd];
            } else {
                a.p[hd].x = ws + L.hid[hd][i - 1];
                a.p[hd].w = W;
                a.p[hd].bias = hd == 0 ? w->add_b[i] : w->rmv_b[i];
            }
            a.p[hd].y = ws + L.hid[hd][i];
            a.p[hd].rows = rows[br];
            a.p[hd].rows_per_inst = rpi[br];
        }
        const int cin = i == 0 ? Cloc : w->head_ch[i - 1];
        a.ldx = cin; a.ldw = w->head_ch[i]; a.K = cin; a.N = w->head_ch[i]; a.relu = 1;
        a.bias_inst_stride = i == 0 ? C0 : 0;
        fill_vec_flags(a, 2);
        rc = launch_layer(a, 2, st);
        if (rc) return rc;
    }
    {
        LrgFinalArgs f = {};
        const int C = w->head_ch[nh - 2];
        if (C % 4 != 0) return LRG_EINVAL - 10;
        for (int hd = 0; hd < 2; ++hd) {
            f.h[hd] = ws + L.hid[hd][nh - 2];
            f.w[hd] = hd == 0 ? w->add_w[nh - 1] : w->rmv_w[nh - 1];
            f.bias[hd] = hd == 0 ? w->add_b[nh - 1] : w->rmv_b[nh - 1];
            f.out[hd] = hd == 0 ? add_logits : rmv_logits;
            f.rows[hd] = rows[hbr[hd]];
        }
        f.C = C;
        long mr = rows[0] > rows[1] ? rows[0] : rows[1];
        hipLaunchKernelGGL(lrg_head_final_kernel, dim3((unsigned)((mr + 15) / 16), 2, 1), dim3(256), 0, st, f);
        LRG_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
