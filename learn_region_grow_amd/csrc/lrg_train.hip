// Training side of LrgNet on gfx950: what the backward pass and the optimiser of
// learn_region_grow_util.py:165-189 need beyond the forward kernels -- a general fp32 matrix-core GEMM with the two
// transposes the gradients take (dX = dZ W^T, dW = X^T dZ over 51 200 rows, split over the reduction), the losses'
// gradient (:165-186), the max-pool's gradient with TensorFlow's tie rule, segment column sums (bias gradients, the
// gradient of the tiled pooled feature) and Adam.  The step itself is sequenced by the host mirror (train.py), as the
// reference sequences it from Python.
#include "lrg_common.h"

typedef float t_f32x16 __attribute__((ext_vector_type(16)));

#define TG_BM 64
#define TG_BN 64
#define TG_BK 32
#define TG_LDA (TG_BK + 4)
#define TG_LDB (TG_BN + 4)

struct LrgGemmArgs {
    const float *A, *B;
    float *C;
    const float *addend;     // nullable [M,N] (ldc): C = acc + addend
    const float *mask;       // nullable [M,N] (ldc): C = mask > 0 ? C : 0   (the ReLU gradient of the layer that PRODUCED the operand)
    int M, N, K, lda, ldb, ldc, transA, transB, kchunk, atomic;
};

// C[M,N] = op(A)[M,K] op(B)[K,N]: one 64x64 tile per 256-thread workgroup (2x2 wavefronts of 32x32, v_mfma_f32_32x32x2_f32),
// K through LDS in chunks of 32; blockIdx.z takes a slice of K and adds its partial tile atomically when the reduction is split.
__global__ __launch_bounds__(256) void lrg_gemm_f32_kernel(LrgGemmArgs a) {
    __shared__ __attribute__((aligned(16))) float As[TG_BM * TG_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[TG_BK * TG_LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * TG_BM, n0 = blockIdx.x * TG_BN;
    const int kbeg = blockIdx.z * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
    t_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += TG_BK) {
        // ---- stage A as As[m][k] ----
        if (!a.transA) {
            for (int idx = tid; idx < TG_BM * TG_BK; idx += 256) {
                const int m = idx >> 5, k = idx & 31;
                float v = 0.f;
                if (m0 + m < a.M && k0 + k < kend) v = a.A[(long)(m0 + m) * a.lda + k0 + k];
                As[m * TG_LDA + k] = v;
            }
        } else {
            for (int idx = tid; idx < TG_BM * TG_BK; idx += 256) {
                const int k = idx >> 6, m = idx & 63;
                float v = 0.f;
                if (m0 + m < a.M && k0 + k < kend) v = a.A[(long)(k0 + k) * a.lda + m0 + m];
                As[m * TG_LDA + k] = v;
            }
        }
        // ---- stage B as Bs[k][n] ----
        if (!a.transB) {
            for (int idx = tid; idx < TG_BK * TG_BN; idx += 256) {
                const int k = idx >> 6, n = idx & 63;
                float v = 0.f;
                if (n0 + n < a.N && k0 + k < kend) v = a.B[(long)(k0 + k) * a.ldb + n0 + n];
                Bs[k * TG_LDB + n] = v;
            }
        } else {
            for (int idx = tid; idx < TG_BK * TG_BN; idx += 256) {
                const int n = idx >> 5, k = idx & 31;
                float v = 0.f;
                if (n0 + n < a.N && k0 + k < kend) v = a.B[(long)(n0 + n) * a.ldb + k0 + k];
                Bs[k * TG_LDB + n] = v;
            }
        }
        __syncthreads();
        const float *ap = &As[(wm * 32 + li) * TG_LDA + 4 * lh];
        const float *bp = &Bs[(4 * lh) * TG_LDB + wn * 32 + li];
#pragma unroll
        for (int g = 0; g < TG_BK / 8; ++g) {                      // lane half h feeds logical k = 8g + 4h + s of both operands
            const float4 av = *reinterpret_cast<const float4 *>(ap + 8 * g);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bp[(8 * g + 0) * TG_LDB], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bp[(8 * g + 1) * TG_LDB], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bp[(8 * g + 2) * TG_LDB], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bp[(8 * g + 3) * TG_LDB], acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + li;
    if (col >= a.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
        if (row < a.M) {
            const long o = (long)row * a.ldc + col;
            if (a.atomic) atomicAdd(&a.C[o], acc[r]);
            else {
                float v = acc[r];
                if (a.addend) v += a.addend[o];
                if (a.mask && !(a.mask[o] > 0.f)) v = 0.f;
                a.C[o] = v;
            }
        }
    }
}

// ---- losses (learn_region_grow_util.py:165-186): d(loss)/d(logits) and the scalars the reference fetches ----
// dlogits[r] = (softmax(logits[r]) - onehot(label[r])) * (label[r] ? w_pos : w_neg).  add head: w_pos = w_neg = 1 / rows
// (:174); remove head: 1 / #positive, 1 / #negative of the batch (:166-172; an empty class contributes nothing).
// stats (double[8], accumulated): 0 sum of weighted ce, 1 argmax == label, 2 true positives, 3 predicted positives,
// 4 labelled positives, 5 rows.
__global__ __launch_bounds__(256) void lrg_ce_grad_kernel(const float *logits, const int32_t *labels, long rows, float w_pos, float w_neg,
                                                           float *dlogits, double *stats) {
    __shared__ double sh[6];
    if (threadIdx.x < 6) sh[threadIdx.x] = 0.0;
    __syncthreads();
    double ce = 0.0;
    int ok = 0, tp = 0, pp = 0, lp = 0, nr = 0;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        const float l0 = logits[2 * r], l1 = logits[2 * r + 1];
        const int y = labels[r] != 0;
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float s = e0 + e1;
        const float p0 = e0 / s, p1 = e1 / s;
        const float w = y ? w_pos : w_neg;
        dlogits[2 * r] = (p0 - (y ? 0.f : 1.f)) * w;
        dlogits[2 * r + 1] = (p1 - (y ? 1.f : 0.f)) * w;
        ce += (double)(w * (logf(s) + m - (y ? l1 : l0)));
        const int pred = l1 > l0;
        ok += pred == y; tp += pred & y; pp += pred; lp += y; ++nr;
    }
    atomicAdd(&sh[0], ce); atomicAdd(&sh[1], (double)ok); atomicAdd(&sh[2], (double)tp);
    atomicAdd(&sh[3], (double)pp); atomicAdd(&sh[4], (double)lp); atomicAdd(&sh[5], (double)nr);
    __syncthreads();
    if (threadIdx.x < 6 && sh[threadIdx.x] != 0.0) atomicAdd(&stats[threadIdx.x], sh[threadIdx.x]);
}

// ---- gradient of the max-pool (:122-123) followed by the pooled layer's ReLU ----
// tf.reduce_max shares the gradient EQUALLY among the rows that tie for the maximum (the copies that pad a small set tie by
// construction); y is post-ReLU, so a zero maximum passes nothing.  One thread per (instance, column): two passes over the rows.
__global__ __launch_bounds__(256) void lrg_pool_backward_kernel(const float *y, const float *dpool, int B, int rows, int C, int dpool_stride,
                                                                 float *dy) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    const float *yb = y + (long)b * rows * C + c;
    float *db = dy + (long)b * rows * C + c;
    float mx = 0.f;
    for (int r = 0; r < rows; ++r) mx = fmaxf(mx, yb[(long)r * C]);
    int ties = 0;
    for (int r = 0; r < rows; ++r) ties += yb[(long)r * C] == mx;
    const float g = mx > 0.f ? dpool[(long)b * dpool_stride + c] / (float)ties : 0.f;
    for (int r = 0; r < rows; ++r) db[(long)r * C] = yb[(long)r * C] == mx ? g : 0.f;
}

// out[s, n] = sum over the seg_rows rows of segment s of x[., n]   (bias gradients; the gradient of the tiled pooled feature)
__global__ __launch_bounds__(256) void lrg_segment_colsum_kernel(const float *x, long n_seg, int seg_rows, int N, float *out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const long s = blockIdx.y;
    if (n >= N) return;
    const float *p = x + s * seg_rows * N + n;
    float acc = 0.f;
    for (int r = 0; r < seg_rows; ++r) acc += p[(long)r * N];
    out[s * N + n] = acc;
}

// ---- tf.compat.v1.train.AdamOptimizer (:188): m, v updated, var -= lr_t * m / (sqrt(v) + eps), lr_t from the host ----
__global__ void lrg_adam_kernel(float *p, const float *g, float *m, float *v, long n, float lr_t, float b1, float b2, float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] + (gi * gi - v[i]) * (1.f - b2);
    m[i] = mi; v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
}

extern "C" {

int lrg_gemm_f32(int M, int N, int K, const float *A, int lda, int transA, const float *B, int ldb, int transB, float *C, int ldc,
                 const float *addend, const float *mask, int split_k, void *stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda <= 0 || ldb <= 0 || ldc < N || split_k < 1) return LRG_EINVAL - 1;
    if (split_k > 1 && (addend || mask)) return LRG_EINVAL - 2;        // a split reduction adds into a zeroed C: no epilogue
    LrgGemmArgs a;
    a.A = A; a.B = B; a.C = C; a.addend = addend; a.mask = mask;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.transA = transA; a.transB = transB;
    const int chunks = (K + TG_BK - 1) / TG_BK;
    const int per = (chunks + split_k - 1) / split_k;
    a.kchunk = per * TG_BK;
    const int nz = (chunks + per - 1) / per;
    a.atomic = split_k > 1;
    hipLaunchKernelGGL(lrg_gemm_f32_kernel, dim3((N + TG_BN - 1) / TG_BN, (M + TG_BM - 1) / TG_BM, nz), dim3(256), 0, (hipStream_t)stream, a);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_ce_grad(const float *logits, const int32_t *labels, long rows, float w_pos, float w_neg, float *dlogits, double *stats,
                void *stream) {
    if (!logits || !labels || !dlogits || !stats || rows <= 0) return LRG_EINVAL - 1;
    const long blocks = (rows + 255) / 256;
    hipLaunchKernelGGL(lrg_ce_grad_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream, logits, labels,
                       rows, w_pos, w_neg, dlogits, stats);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_pool_backward(const float *y, const float *dpool, int B, int rows, int C, int dpool_stride, float *dy, void *stream) {
    if (!y || !dpool || !dy || B <= 0 || rows <= 0 || C <= 0 || dpool_stride < C) return LRG_EINVAL - 1;
    hipLaunchKernelGGL(lrg_pool_backward_kernel, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, y, dpool, B, rows, C,
                       dpool_stride, dy);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_segment_colsum(const float *x, long n_seg, int seg_rows, int N, float *out, void *stream) {
    if (!x || !out || n_seg <= 0 || seg_rows <= 0 || N <= 0 || n_seg > 65535) return LRG_EINVAL - 1;
    hipLaunchKernelGGL(lrg_segment_colsum_kernel, dim3((N + 255) / 256, (unsigned)n_seg), dim3(256), 0, (hipStream_t)stream, x, n_seg,
                       seg_rows, N, out);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_adam_step(float *params, const float *grads, float *m, float *v, long n, float lr_t, float beta1, float beta2, float epsilon,
                  void *stream) {
    if (!params || !grads || !m || !v || n <= 0) return LRG_EINVAL - 1;
    hipLaunchKernelGGL(lrg_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n, lr_t,
                       beta1, beta2, epsilon);
    LRG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
