// Room preprocessing P0 on the GPU (test_region_grow.py:119-173): first-point-per-voxel equalisation, per-point PCA over
// the raw points of the 27 surrounding voxels, the 13-column feature stack.
//
// Everything up to and including the covariance matrix is computed in the reference's own arithmetic AND order -- voxel
// of a point = rint(float32 x / float32 res); neighbours visited in itertools.product([-1,0,1]^3) order, raw points in
// file order inside a voxel; float32 outer products accumulated in float64; cov = accA/n - outer(accB,accB)/n^2 -- so the
// covariances are bit-identical to the NumPy loop.  The 3x3 decomposition is where a GPU cannot follow the reference
// (LAPACK dgesdd inside numpy.linalg.svd): eig_mode 0 stops after the covariances (the host runs the same LAPACK call:
// bit-exact features), eig_mode 1 solves them here with a cyclic Jacobi iteration in float64 (|error| ~ 1e-16 |cov|:
// features agree to float32 rounding, documented tolerance in tests/test_gpu_preprocess.py).
#include "lrg_common.h"

#define PREP_THREADS 256
#define PREP_SCAN_ITEMS 8                      // per thread: 2048 elements per block

struct LrgPrepLayout {
    size_t keys, first, count, off, rank, cursor;     // per hash slot
    size_t slot, flag, list;                          // per raw point
    size_t bsum;                                      // block sums of the scans
    size_t scal;                                      // scalars: [0] N, [1] error, [2..7] xyz min/max (ordered ints), [8,9] max curvature (u64), [10] any-NaN
    size_t normal, curv;                              // per equalised point: float64 [3], float64
    size_t total;
    int cap;
};

static int prep_layout(int M, LrgPrepLayout *L) {
    if (M <= 0) return LRG_EINVAL - 50;
    long cap = 64;
    while (cap < 2L * M) cap <<= 1;
    if (cap > (1L << 30)) return LRG_EINVAL - 51;
    L->cap = (int)cap;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = lrg_align_up(o + bytes, 256); return at; };
    L->keys = take((size_t)cap * 8);
    L->first = take((size_t)cap * 4);
    L->count = take((size_t)cap * 4);
    L->off = take((size_t)cap * 4);
    L->rank = take((size_t)cap * 4);
    L->cursor = take((size_t)cap * 4);
    L->slot = take((size_t)M * 4);
    L->flag = take((size_t)M * 4);
    L->list = take((size_t)M * 4);
    const long per_block = PREP_THREADS * PREP_SCAN_ITEMS;
    const long nb = ((cap > M ? cap : M) + per_block - 1) / per_block;
    L->bsum = take((size_t)(nb + 1) * 4);
    L->scal = take(64 * 4);
    L->normal = take((size_t)M * 3 * 8);
    L->curv = take((size_t)M * 8);
    L->total = o;
    return 0;
}

// ---- order-preserving integer images of floats / doubles (for atomicMin / atomicMax) ----
__device__ __forceinline__ int prep_ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float prep_unord(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void prep_init_kernel(uint64_t *keys, int32_t *first, int32_t *count, int32_t *cursor, int cap, int32_t *scal) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) { keys[i] = LRG_HASH_EMPTY; first[i] = INT_MAX; count[i] = 0; cursor[i] = 0; }
    if (i < 64) scal[i] = (i >= 2 && i <= 4) ? INT_MAX : (i >= 5 && i <= 7) ? INT_MIN : 0;
}

// voxel -> slot (insert), first raw index and population of every voxel          (:125-133)
__global__ void prep_insert_kernel(const float *raw, int ld, int M, float res, uint64_t *keys, int32_t *first, int32_t *count,
                                   int mask, int32_t *slot_of, int32_t *scal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float *p = raw + (long)i * ld;
    const uint64_t key = lrg_pack_voxel(lrg_voxel_of(p[0], res), lrg_voxel_of(p[1], res), lrg_voxel_of(p[2], res));
    if (key == LRG_HASH_EMPTY) { scal[1] = 1; slot_of[i] = -1; return; }     // outside the 21-bit voxel window
    unsigned h = (unsigned)lrg_fmix64(key) & (unsigned)mask;
    while (true) {
        unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long *>(&keys[h]), (unsigned long long)LRG_HASH_EMPTY,
                                            (unsigned long long)key);
        if (prev == LRG_HASH_EMPTY || prev == key) break;
        h = (h + 1) & (unsigned)mask;
    }
    atomicMin(&first[h], i);
    atomicAdd(&count[h], 1);
    slot_of[i] = (int)h;
}

__global__ void prep_flag_kernel(const int32_t *slot_of, const int32_t *first, int M, int32_t *flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) flag[i] = (slot_of[i] >= 0 && first[slot_of[i]] == i) ? 1 : 0;
}

// ---- exclusive scan of an int32 array (three launches; block = 2048 elements) ----
__device__ __forceinline__ int prep_block_exscan(int v, int *sh, int *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int k = 0; k < PREP_THREADS / 64; ++k) {
        if (k < w) base += sh[k];
        tot += sh[k];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(PREP_THREADS) void prep_scan_sums_kernel(const int32_t *x, long n, int32_t *bsum) {
    __shared__ int sh[PREP_THREADS / 64];
    const long base = ((long)blockIdx.x * PREP_THREADS + threadIdx.x) * PREP_SCAN_ITEMS;
    int s = 0;
    for (int k = 0; k < PREP_SCAN_ITEMS; ++k) s += base + k < n ? x[base + k] : 0;
    int tot;
    prep_block_exscan(s, sh, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(PREP_THREADS) void prep_scan_top_kernel(int32_t *bsum, int nb) {   // one block; bsum[nb] = total
    __shared__ int sh[PREP_THREADS / 64];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += PREP_THREADS) {
        const int i = b0 + threadIdx.x;
        const int v = i < nb ? bsum[i] : 0;
        int tot;
        const int ex = prep_block_exscan(v, sh, &tot);
        if (i < nb) bsum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) bsum[nb] = carry;
}

__global__ __launch_bounds__(PREP_THREADS) void prep_scan_apply_kernel(const int32_t *x, long n, const int32_t *bsum, int32_t *out) {
    __shared__ int sh[PREP_THREADS / 64];
    const long base = ((long)blockIdx.x * PREP_THREADS + threadIdx.x) * PREP_SCAN_ITEMS;
    int v[PREP_SCAN_ITEMS], s = 0;
    for (int k = 0; k < PREP_SCAN_ITEMS; ++k) { v[k] = base + k < n ? x[base + k] : 0; s += v[k]; }
    int tot;
    int run = bsum[blockIdx.x] + prep_block_exscan(s, sh, &tot);
    for (int k = 0; k < PREP_SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

static int prep_exscan(const int32_t *x, long n, int32_t *bsum, int32_t *out, hipStream_t st) {
    const long per_block = PREP_THREADS * PREP_SCAN_ITEMS;
    const int nb = (int)((n + per_block - 1) / per_block);
    hipLaunchKernelGGL(prep_scan_sums_kernel, dim3(nb), dim3(PREP_THREADS), 0, st, x, n, bsum);
    hipLaunchKernelGGL(prep_scan_top_kernel, dim3(1), dim3(PREP_THREADS), 0, st, bsum, nb);
    hipLaunchKernelGGL(prep_scan_apply_kernel, dim3(nb), dim3(PREP_THREADS), 0, st, x, n, bsum, out);
    LRG_LAUNCH_CHECK();
    return 0;
}

// equalised order = raw order of the first point of each voxel (:127-129,:134); every raw point learns its voxel's rank (:130)
__global__ void prep_equalize_kernel(const int32_t *flag, const int32_t *rank_of_raw, const int32_t *slot_of, int M,
                                     int32_t *equalized_idx, int32_t *hash_rank, const int32_t *bsum_total, int32_t *scal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) scal[0] = *bsum_total;
    if (i >= M || !flag[i]) return;
    equalized_idx[rank_of_raw[i]] = i;
    hash_rank[slot_of[i]] = rank_of_raw[i];
}

__global__ void prep_fill_kernel(const int32_t *slot_of, const int32_t *hash_off, const int32_t *hash_rank, int32_t *cursor, int M,
                                 int32_t *list, int32_t *unequalized_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int s = slot_of[i];
    if (s < 0) { unequalized_idx[i] = -1; return; }
    list[hash_off[s] + atomicAdd(&cursor[s], 1)] = i;
    unequalized_idx[i] = hash_rank[s];
}

// raw points of a voxel in file order (normal_grid[k].append(i), :131-133)
__global__ void prep_sort_lists_kernel(const int32_t *hash_off, const int32_t *count, int cap, int32_t *list) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap) return;
    const int n = count[s];
    if (n < 2) return;
    int32_t *a = list + hash_off[s];
    for (int i = 1; i < n; ++i) {
        const int v = a[i];
        int j = i - 1;
        while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; }
        a[j + 1] = v;
    }
}

#define PREP_EIG_SLACK (256.0 * 2.220446049250313e-16)
// Eigen-decomposition of a symmetric 3x3 matrix, cyclic Jacobi in float64.  w: eigenvalues, V[k][:]: eigenvector k.
__device__ void prep_jacobi3(const double *c, double *w, double (*V)[3]) {
    double a[3][3] = {{c[0], c[1], c[2]}, {c[1], c[4], c[5]}, {c[2], c[5], c[8]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};             // columns = eigenvectors
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off == 0.0 || off <= 1e-300 || off < 1e-22 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = a[p][q];
                if (apq == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                const int r = 3 - p - q;
                const double app = a[p][p], aqq = a[q][q], arp = a[r][p], arq = a[r][q];
                a[p][p] = app - t * apq;
                a[q][q] = aqq + t * apq;
                a[p][q] = a[q][p] = 0.0;
                a[r][p] = a[p][r] = cs * arp - sn * arq;
                a[r][q] = a[q][r] = sn * arp + cs * arq;
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = cs * vkp - sn * vkq;
                    v[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    for (int k = 0; k < 3; ++k) { w[k] = a[k][k]; V[k][0] = v[0][k]; V[k][1] = v[1][k]; V[k][2] = v[2][k]; }
}

// covariance of the raw points in the 27 voxels around every equalised point (:144-157), then normal and curvature (:158-161)
__global__ __launch_bounds__(PREP_THREADS) void prep_cov_kernel(const float *raw, int ld, float res, const int32_t *equalized_idx,
                                                              const int32_t *scal_n, const uint64_t *keys, const int32_t *hash_off,
                                                              const int32_t *count, int mask, const int32_t *list, double *cov_out,
                                                              int eig_mode, double *normal, double *curv, int32_t *scal, int32_t *nflag) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= *scal_n) return;
    const float *pe = raw + (long)equalized_idx[e] * ld;
    const int vx = lrg_voxel_of(pe[0], res), vy = lrg_voxel_of(pe[1], res), vz = lrg_voxel_of(pe[2], res);
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, B[3] = {0, 0, 0};
    int n = 0;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {                                 // itertools.product order (:147)
                const uint64_t key = lrg_pack_voxel(vx + dx, vy + dy, vz + dz);
                if (key == LRG_HASH_EMPTY) continue;
                unsigned h = (unsigned)lrg_fmix64(key) & (unsigned)mask;
                int slot = -1;
                while (true) {
                    const uint64_t k = keys[h];
                    if (k == key) { slot = (int)h; break; }
                    if (k == LRG_HASH_EMPTY) break;
                    h = (h + 1) & (unsigned)mask;
                }
                if (slot < 0) continue;
                const int o = hash_off[slot], c = count[slot];
                for (int t = 0; t < c; ++t) {
                    const float *p = raw + (long)list[o + t] * ld;
                    const float x = p[0], y = p[1], z = p[2];
                    // numpy.outer(p, p) of a float32 row is float32; += into the float64 accumulator (:155)
                    A[0] += (double)__fmul_rn(x, x); A[1] += (double)__fmul_rn(x, y); A[2] += (double)__fmul_rn(x, z);
                    A[3] += (double)__fmul_rn(y, x); A[4] += (double)__fmul_rn(y, y); A[5] += (double)__fmul_rn(y, z);
                    A[6] += (double)__fmul_rn(z, x); A[7] += (double)__fmul_rn(z, y); A[8] += (double)__fmul_rn(z, z);
                    B[0] += (double)x; B[1] += (double)y; B[2] += (double)z;    // (:156)
                }
                n += c;
            }
    const double dn = (double)n, dn2 = dn * dn;
    double C[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i + j] / dn - (B[i] * B[j]) / dn2;          // (:157)
    if (cov_out)
        for (int k = 0; k < 9; ++k) cov_out[(long)e * 9 + k] = C[k];
    // room extent for the normalised coordinates (:139)
    atomicMin(&scal[2], prep_ord(pe[0])); atomicMin(&scal[3], prep_ord(pe[1])); atomicMin(&scal[4], prep_ord(pe[2]));
    atomicMax(&scal[5], prep_ord(pe[0])); atomicMax(&scal[6], prep_ord(pe[1])); atomicMax(&scal[7], prep_ord(pe[2]));
    if (!eig_mode) return;
    double w[3], V[3][3];
    prep_jacobi3(C, w, V);
    // singular values of a symmetric matrix = |eigenvalues|, descending; V[2] belongs to the smallest (:158-159)
    double s[3] = {fabs(w[0]), fabs(w[1]), fabs(w[2])};
    int i0 = 0, i2 = 0;
    if (s[1] > s[i0]) i0 = 1;
    if (s[2] > s[i0]) i0 = 2;
    if (s[1] < s[i2]) i2 = 1;
    if (s[2] <= s[i2]) i2 = 2;
    if (i0 == i2) { i0 = 0; i2 = 2; }                                                               // all equal
    const int i1 = 3 - i0 - i2;
    normal[(long)e * 3 + 0] = fabs(V[i2][0]); normal[(long)e * 3 + 1] = fabs(V[i2][1]); normal[(long)e * 3 + 2] = fabs(V[i2][2]);
    const double cv = fabs(s[i2] / (s[i0] + s[i1] + s[i2]));                                         // S[2]/(S[0]+S[1]+S[2]) (:160-161)
    curv[e] = cv;
    if (eig_mode == 2 && nflag) {
        // Would LAPACK's decomposition of the same matrix round to the same float32 normal?  Both solvers are backward stable: their
        // eigenvectors of the smallest eigenvalue differ by at most ~p eps |A| / (gap to the next eigenvalue) with a small p; with
        // PREP_EIG_SLACK = 256 eps (an order of magnitude above either solver's constant) a component whose float32 rounding is the
        // same at both ends of that interval is the same float32 number under LAPACK.  Everything else -- near-degenerate pairs of
        // eigenvalues, components next to a rounding boundary, NaN -- is flagged and redone by the host's LAPACK call (preprocess_gpu).
        const double gap = s[i1] - s[i2];
        int unsafe = !(gap > 1e-6 * s[i0]) || !(cv == cv);
        if (!unsafe) {
            const double dv = PREP_EIG_SLACK * s[i0] / gap;
            for (int k = 0; k < 3; ++k) {
                const double x = fabs(V[i2][k]);
                if ((float)(x - dv) != (float)(x + dv) || x < dv) unsafe = 1;
            }
        }
        nflag[e] = unsafe;
    }
    if (cv != cv) scal[10] = 1;                                                                     // numpy's max() propagates NaN
    else atomicMax(reinterpret_cast<unsigned long long *>(&scal[8]), (unsigned long long)__double_as_longlong(cv));
}

// the feature stack (:163-172)
__global__ void prep_features_kernel(const float *raw, int ld, const int32_t *obj, const int32_t *cls, const int32_t *equalized_idx,
                                     const int32_t *scal, const double *normal, double *curv, int F, float *points, int32_t *obj_out,
                                     int32_t *cls_out, int keep_raw_curv) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= scal[0]) return;
    const int i = equalized_idx[e];
    const float *p = raw + (long)i * ld;
    float *o = points + (long)e * F;
    for (int k = 0; k < 3; ++k) {
        const float mn = prep_unord(scal[2 + k]), mx = prep_unord(scal[5 + k]);
        o[k] = p[k];
        o[3 + k] = __fdiv_rn(__fsub_rn(p[k], mn), __fsub_rn(mx, mn));
    }
    if (F >= 9)
        for (int k = 0; k < 3; ++k) o[6 + k] = p[3 + k];
    if (F >= 12)
        for (int k = 0; k < 3; ++k) o[9 + k] = (float)normal[(long)e * 3 + k];
    const double cmax = scal[10] ? __longlong_as_double(0x7ff8000000000000LL)
                                 : __longlong_as_double(*reinterpret_cast<const long long *>(&scal[8]));
    const double c = curv[e] / cmax;                                                                 // (:163)
    if (!keep_raw_curv) curv[e] = c;      // (eig_mode 2: the host normalises with LAPACK's own maximum)
    if (F >= 13) o[12] = (float)c;
    if (obj_out) obj_out[e] = obj ? obj[i] : 0;
    if (cls_out) cls_out[e] = cls ? cls[i] : 0;
}

extern "C" {

size_t lrg_preprocess_workspace_bytes(int n_raw) {
    LrgPrepLayout L;
    if (prep_layout(n_raw, &L) != 0) return 0;
    return L.total;
}

int lrg_preprocess(const float *raw, int raw_stride, const int32_t *obj_id, const int32_t *cls_id, int n_raw, float resolution,
                   int feature_size, int eig_mode, void *workspace, size_t workspace_bytes, float *points, int32_t *obj_out,
                   int32_t *cls_out, double *curvatures, int32_t *equalized_idx, int32_t *unequalized_idx, double *cov,
                   int32_t *n_equalized, void *stream) {
    LrgPrepLayout L;
    int rc = prep_layout(n_raw, &L);
    if (rc) return rc;
    if (!raw || raw_stride < 6 || !workspace || !equalized_idx || !unequalized_idx || !n_equalized) return LRG_EINVAL - 52;
    if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return LRG_EINVAL - 53;
    if (!(resolution > 0.f)) return LRG_EINVAL - 54;
    if (feature_size != 6 && feature_size != 9 && feature_size != 12 && feature_size != 13) return LRG_EINVAL - 55;
    if (eig_mode != 0 && eig_mode != 1 && eig_mode != 2) return LRG_EINVAL - 56;
    if (eig_mode >= 1 && (!points || !curvatures)) return LRG_EINVAL - 57;
    if ((eig_mode == 0 || eig_mode == 2) && !cov) return LRG_EINVAL - 58;
    hipStream_t st = (hipStream_t)stream;
    char *ws = static_cast<char *>(workspace);
    uint64_t *keys = reinterpret_cast<uint64_t *>(ws + L.keys);
    int32_t *first = reinterpret_cast<int32_t *>(ws + L.first), *count = reinterpret_cast<int32_t *>(ws + L.count);
    int32_t *hoff = reinterpret_cast<int32_t *>(ws + L.off), *hrank = reinterpret_cast<int32_t *>(ws + L.rank);
    int32_t *cursor = reinterpret_cast<int32_t *>(ws + L.cursor), *slot = reinterpret_cast<int32_t *>(ws + L.slot);
    int32_t *flag = reinterpret_cast<int32_t *>(ws + L.flag), *list = reinterpret_cast<int32_t *>(ws + L.list);
    int32_t *bsum = reinterpret_cast<int32_t *>(ws + L.bsum), *scal = reinterpret_cast<int32_t *>(ws + L.scal);
    double *normal = reinterpret_cast<double *>(ws + L.normal);
    double *curv = curvatures ? curvatures : reinterpret_cast<double *>(ws + L.curv);
    const int cap = L.cap, mask = cap - 1;
    const int gm = (n_raw + PREP_THREADS - 1) / PREP_THREADS, gc = (cap + PREP_THREADS - 1) / PREP_THREADS;
    hipLaunchKernelGGL(prep_init_kernel, dim3(gc), dim3(PREP_THREADS), 0, st, keys, first, count, cursor, cap, scal);
    hipLaunchKernelGGL(prep_insert_kernel, dim3(gm), dim3(PREP_THREADS), 0, st, raw, raw_stride, n_raw, resolution, keys, first, count,
                       mask, slot, scal);
    hipLaunchKernelGGL(prep_flag_kernel, dim3(gm), dim3(PREP_THREADS), 0, st, slot, first, n_raw, flag);
    LRG_LAUNCH_CHECK();
    // rank of every first point (in place over the flags' scan output: the flag is needed afterwards, so scan into `list`)
    if ((rc = prep_exscan(flag, n_raw, bsum, list, st))) return rc;
    const long per_block = PREP_THREADS * PREP_SCAN_ITEMS;
    const int nb_m = (int)((n_raw + per_block - 1) / per_block);
    hipLaunchKernelGGL(prep_equalize_kernel, dim3(gm), dim3(PREP_THREADS), 0, st, flag, list, slot, n_raw, equalized_idx, hrank,
                       bsum + nb_m, scal);
    LRG_HIP_CHECK(hipMemcpyAsync(n_equalized, scal, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if ((rc = prep_exscan(count, cap, bsum, hoff, st))) return rc;
    hipLaunchKernelGGL(prep_fill_kernel, dim3(gm), dim3(PREP_THREADS), 0, st, slot, hoff, hrank, cursor, n_raw, list, unequalized_idx);
    hipLaunchKernelGGL(prep_sort_lists_kernel, dim3(gc), dim3(PREP_THREADS), 0, st, hoff, count, cap, list);
    hipLaunchKernelGGL(prep_cov_kernel, dim3(gm), dim3(PREP_THREADS), 0, st, raw, raw_stride, resolution, equalized_idx, scal, keys,
                       hoff, count, mask, list, cov, eig_mode, normal, curv, scal, flag);
    if (eig_mode >= 1)
        hipLaunchKernelGGL(prep_features_kernel, dim3(gm), dim3(PREP_THREADS), 0, st, raw, raw_stride, obj_id, cls_id, equalized_idx, scal,
                           normal, curv, feature_size, points, obj_out, cls_out, eig_mode == 2 ? 1 : 0);
    LRG_LAUNCH_CHECK();
    return 0;
}

/* eig_mode 2: which equalised points' float32 normals are not certain to equal LAPACK's (1) -- copied out of the workspace */
int lrg_preprocess_unsafe_normals(const void *workspace, int n_raw, int n_equalized, int32_t *flags_out, void *stream) {
    LrgPrepLayout L;
    int rc = prep_layout(n_raw, &L);
    if (rc) return rc;
    if (!workspace || !flags_out || n_equalized < 0 || n_equalized > n_raw) return LRG_EINVAL - 52;
    if (n_equalized == 0) return 0;
    LRG_HIP_CHECK(hipMemcpyAsync(flags_out, static_cast<const char *>(workspace) + L.flag, (size_t)n_equalized * sizeof(int32_t), hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
    return 0;
}

/* error flag of the last lrg_preprocess on this workspace: 1 = a point fell outside the 21-bit voxel window */
int lrg_preprocess_status(const void *workspace, int n_raw, int32_t *host_status, void *stream) {
    LrgPrepLayout L;
    int rc = prep_layout(n_raw, &L);
    if (rc) return rc;
    if (!workspace || !host_status) return LRG_EINVAL - 52;
    LRG_HIP_CHECK(hipMemcpyAsync(host_status, static_cast<const char *>(workspace) + L.scal + 4, sizeof(int32_t), hipMemcpyDeviceToHost,
                                 (hipStream_t)stream));
    LRG_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

}  // extern "C"
