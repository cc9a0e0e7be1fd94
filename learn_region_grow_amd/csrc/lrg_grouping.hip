// tf_ops/grouping replacements for gfx950 (reference: tf_ops/grouping/tf_grouping_g.cu).
// The reference launches <<<b,256>>> -- one block per batch item, one THREAD per query row, serial over n.
// Here one 64-lane wavefront owns a query / a row and scans n with ballots, so a launch fills the chip.
#include "lrg_common.h"

// ---- ball query (tf_grouping_g.cu:3-36): FIRST nsample points with max(sqrt(d2),1e-20) < radius ----
__global__ __launch_bounds__(256) void lrg_query_ball_kernel(int b, int n, int m, float radius, int nsample,
                                                              const float *xyz1, const float *xyz2, int *idx,
                                                              int *pts_cnt) {
    const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);      // one wave per query
    if (q >= (long)b * m) return;
    const int lane = lrg_lane();
    const long bi = q / m;
    const float *p1 = xyz1 + bi * n * 3;
    const float x2 = xyz2[q * 3 + 0], y2 = xyz2[q * 3 + 1], z2 = xyz2[q * 3 + 2];
    int *out = idx + q * nsample;
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    int cnt = 0, first = -1;
    for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
        int k = k0 + lane;
        bool hit = false;
        if (k < n) {
            float dx = __fsub_rn(x2, p1[k * 3 + 0]), dy = __fsub_rn(y2, p1[k * 3 + 1]), dz = __fsub_rn(z2, p1[k * 3 + 2]);
            float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            float d = fmaxf(__fsqrt_rn(d2), 1e-20f);
            hit = d < radius;
        }
        unsigned long long mask = __ballot(hit);
        if (mask) {
            if (first < 0) first = k0 + (int)__ffsll((long long)mask) - 1;
            int pos = cnt + __popcll(mask & lt);
            if (hit && pos < nsample) out[pos] = k;
            cnt += __popcll(mask);
        }
    }
    if (cnt > nsample) cnt = nsample;
    // slots never reached repeat the first hit (:26-29); rows with no hit are zero-filled
    for (int l = cnt + lane; l < nsample; l += 64) out[l] = first < 0 ? 0 : first;
    if (lane == 0) pts_cnt[q] = cnt;
}

// ---- gather rows (tf_grouping_g.cu:40-57) ----
__global__ void lrg_group_point_kernel(long total, int n, int c, int m, int nsample, const float *points, const int *idx,
                                       float *out) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    long g = e / c;                 // (b, j, k) flattened
    int l = (int)(e - g * c);
    long bi = g / ((long)m * nsample);
    int ii = idx[g];
    out[e] = points[(bi * n + ii) * c + l];
}

// ---- scatter-add gradient (tf_grouping_g.cu:61-78) ----
__global__ void lrg_group_point_grad_kernel(long total, int n, int c, int m, int nsample, const float *grad_out,
                                            const int *idx, float *grad_points) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    long g = e / c;
    int l = (int)(e - g * c);
    long bi = g / ((long)m * nsample);
    int ii = idx[g];
    atomicAdd(&grad_points[(bi * n + ii) * c + l], grad_out[e]);
}

// ---- partial selection sort (tf_grouping_g.cu:83-123), one wave per row, exact swap sequence ----
__global__ __launch_bounds__(256) void lrg_selection_sort_kernel(long rows, int n, int k, const float *dist, int *outi,
                                                                  float *out) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = lrg_lane();
    const float *d = dist + r * n;
    float *o = out + r * n;
    int *oi = outi + r * n;
    for (int s = lane; s < n; s += 64) { o[s] = d[s]; oi[s] = s; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int s = 0; s < k && s < n; ++s) {
        // first minimum over t in [s, n): strict '<' keeps the earliest position on ties
        float best = INFINITY; int bt = INT_MAX;
        for (int t = s + lane; t < n; t += 64) {
            float v = o[t];
            if (bt == INT_MAX || v < best) { best = v; bt = t; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            float ov = __shfl_xor(best, off); int ot = __shfl_xor(bt, off);
            if (ot != INT_MAX && (bt == INT_MAX || ov < best || (ov == best && ot < bt))) { best = ov; bt = ot; }
        }
        if (lane == 0 && bt != s) {
            float tv = o[bt]; o[bt] = o[s]; o[s] = tv;
            int ti = oi[bt]; oi[bt] = oi[s]; oi[s] = ti;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- squared-distance matrix of knn_point (tf_grouping.py:62-65) ----
__global__ void lrg_pairwise_sqdist_kernel(int b, int n, int m, int c, const float *xyz1, const float *xyz2,
                                           float *dist) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)b * m * n;
    if (e >= total) return;
    int k = (int)(e % n);
    long bj = e / n;
    long bi = bj / m;
    const float *p1 = xyz1 + (bi * n + k) * c;
    const float *p2 = xyz2 + bj * c;
    float s = 0.f;
    for (int l = 0; l < c; ++l) {
        float d = __fsub_rn(p1[l], p2[l]);
        float sq = __fmul_rn(d, d);
        s = l == 0 ? sq : __fadd_rn(s, sq);
    }
    dist[e] = s;
}

extern "C" {

int lrg_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2, int *idx,
                         int *pts_cnt, void *stream) {
    if (b < 0 || n < 0 || m < 0 || nsample <= 0 || !xyz1 || !xyz2 || !idx || !pts_cnt) return LRG_EINVAL - 1;
    long q = (long)b * m;
    if (q == 0) return 0;
    hipLaunchKernelGGL(lrg_query_ball_kernel, dim3((unsigned)((q + 3) / 4)), dim3(256), 0, (hipStream_t)stream, b, n, m,
                       radius, nsample, xyz1, xyz2, idx, pts_cnt);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream) {
    if (b < 0 || n <= 0 || m < 0 || k < 0 || !dist || !outi || !out) return LRG_EINVAL - 1;
    long rows = (long)b * m;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(lrg_selection_sort_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       rows, n, k, dist, outi, out);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out,
                    void *stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || !points || !idx || !out) return LRG_EINVAL - 1;
    long total = (long)b * m * nsample * c;
    if (total == 0) return 0;
    hipLaunchKernelGGL(lrg_group_point_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       total, n, c, m, nsample, points, idx, out);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                         float *grad_points, void *stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || !grad_out || !idx || !grad_points) return LRG_EINVAL - 1;
    long total = (long)b * m * nsample * c;
    if (total == 0) return 0;
    hipLaunchKernelGGL(lrg_group_point_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, total, n, c, m, nsample, grad_out, idx, grad_points);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_pairwise_sqdist(int b, int n, int m, int c, const float *xyz1, const float *xyz2, float *dist, void *stream) {
    if (b < 0 || n < 0 || m < 0 || c <= 0 || !xyz1 || !xyz2 || !dist) return LRG_EINVAL - 1;
    long total = (long)b * m * n;
    if (total == 0) return 0;
    hipLaunchKernelGGL(lrg_pairwise_sqdist_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, b, n, m, c, xyz1, xyz2, dist);
    LRG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
