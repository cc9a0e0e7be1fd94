// tf_ops/grouping replacements for gfx950 (reference: tf_ops/grouping/tf_grouping_g.cu).
// The reference launches <<<b,256>>> -- one block per batch item, one THREAD per query row, serial over n.
// Here one 64-lane wavefront owns a query / a row and scans n with ballots, so a launch fills the chip.
#include "lrg_common.h"
#include <type_traits>

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
    // Four chunks of 64 points per round, their twelve loads issued together (the scan's early exit made every chunk wait for the one before it: eight
    // dependent trips to L2 for n = 512, 12 us at the harness shape); chunks behind the one that fills the list are skipped as before.
    constexpr int UN = 4;
    for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64 * UN) {
        float px[UN], py[UN], pz[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = min(k0 + 64 * u + lane, n - 1);
            px[u] = p1[k * 3 + 0]; py[u] = p1[k * 3 + 1]; pz[u] = p1[k * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = k0 + 64 * u + lane;
            if (k0 + 64 * u >= n || cnt >= nsample) break;
            bool hit = false;
            if (k < n) {
                float dx = __fsub_rn(x2, px[u]), dy = __fsub_rn(y2, py[u]), dz = __fsub_rn(z2, pz[u]);
                float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                float d = fmaxf(__fsqrt_rn(d2), 1e-20f);
                hit = d < radius;
            }
            unsigned long long mask = __ballot(hit);
            if (mask) {
                if (first < 0) first = k0 + 64 * u + (int)__ffsll((long long)mask) - 1;
                int pos = cnt + __popcll(mask & lt);
                if (hit && pos < nsample) out[pos] = k;
                cnt += __popcll(mask);
            }
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

// the same gather moving 16 bytes per thread with 32-bit index arithmetic (c a multiple of 4, fewer than 2^31 elements)
#ifndef LRG_GP_U
#define LRG_GP_U 2
#endif
typedef float lrg_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void lrg_group_point_vec4_kernel(unsigned total4, int n, int c4, int m_ns, const float4 *points, const int *idx, float4 *out) {
    // LRG_GP_U pieces of 16 bytes per thread, a workgroup's pieces contiguous per step: the index and row loads of all of them in flight, then the stores
    // (written once, never read here: non-temporal).  Harness shape, kernel time under rocprofv3: one piece per thread and plain stores 15.1 us, 2 / 4 / 8 pieces
    // 11.1 / 12.0 / 13.2 us = 0.81 of the HBM peak for the 72 MB that have to move.
    const unsigned e0 = blockIdx.x * (256u * LRG_GP_U) + threadIdx.x;
    lrg_v4f v[LRG_GP_U];
#pragma unroll
    for (int u = 0; u < LRG_GP_U; ++u) {
        const unsigned e = min(e0 + 256u * u, total4 - 1);
        const unsigned g = e / (unsigned)c4;
        const unsigned l = e - g * (unsigned)c4;
        const unsigned bi = g / (unsigned)m_ns;
        v[u] = *reinterpret_cast<const lrg_v4f *>(&points[((size_t)bi * n + idx[g]) * c4 + l]);
    }
#pragma unroll
    for (int u = 0; u < LRG_GP_U; ++u) {
        const unsigned e = e0 + 256u * u;
        if (e < total4) __builtin_nontemporal_store(v[u], reinterpret_cast<lrg_v4f *>(out) + e);
    }
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

// The same scatter-add without a single global atomic: a workgroup OWNS the gradient of (batch item, 16 channels) -- n x 16 floats in LDS.
// The reference's kernel -- and the one above -- send every element to memory as an atomic of its own: b x m x nsample rows of c floats
// onto b x n target rows, and the ball query pads its index lists with copies of the first hit (tf_grouping_g.cu:20-23), so most atomics
// of a query land on the same few rows and serialise there (203 us for 67 MB of gradients = 0.045 of the HBM peak,
// profiles/r03_grouping_rates.json; pre-reduced per (batch item, row range) in LDS and flushed with one atomic per target element: 97 us --
// the flush is still millions of atomics).  Here the workgroup streams its slice (32 bytes since the end of round 4) of every gradient row of the batch item
// (16 bytes per lane, the next rows' loads in flight), adds it into LDS (ds_add_f32: nothing leaves the CU) and stores its n x 16 sums with plain
// stores.  Every input byte is read once, every output byte written once; the order of the additions differs from launch to launch as
// the reference's does (atomicAdd, :61-78).
#ifndef LRG_GPG_SLICE
#define LRG_GPG_SLICE 8
#endif
#ifndef LRG_GPG_U
#define LRG_GPG_U 2
#endif
// SLICE channels per workgroup (16: four 16-byte pieces per gradient row, 8: two).  Harness shape (b = 32, n = 512, m x nsample = 8 192, c = 64; profiles/r04_g4_ab.txt):
// 16 channels = 128 workgroups on 256 CUs, one buffer of 8 rows per lane: 37 us (0.245 of the HBM peak); two buffers: 32-33 us; 8 channels = 256 workgroups: 22.0 us
// (0.41; two rows per lane and buffer; 4 / 8 rows: 22.6 / 24.8 us); 4 channels = 512 workgroups: 29-31 us (a wavefront's load touches 64 rows for 16 bytes each).  The workgroups of one batch item sit on ONE XCD (workgroup id -> XCD round robin:
// id = 8 k + x runs on XCD x), so that the 64-byte sectors of a 256-byte gradient row that its slices fetch come through the same L2.  Two buffers of U rows per
// lane: the next pass's loads are in flight while this pass's rows are combined and added.
template <int SLICE, int U>
__global__ __launch_bounds__(512) void lrg_group_point_grad_lds_kernel(int b, int n, int c, int m_ns, const float *grad_out, const int *idx,
                                                                       float *grad_points, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float gp_acc[];      // [n][SLICE]
    constexpr int L = SLICE / 4;                                        // 16-byte pieces (lanes) per row
    constexpr int WROWS = 64 / L, PASS = 8 * WROWS;                     // rows per wavefront / per workgroup and load
    const int tid = threadIdx.x;
    const int nslice = c / SLICE, G = 8 * nslice, id_in = (int)blockIdx.x % G;
    const int bi = ((int)blockIdx.x / G) * 8 + (id_in & 7), ch0 = (id_in >> 3) * SLICE;
    if (bi >= b) return;
    const int total = n * SLICE;
    for (int e = 4 * tid; e < total; e += 4 * 512) *reinterpret_cast<float4 *>(gp_acc + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // A wavefront holds WROWS consecutive rows; lane = 16 * g + ri: the 16 lanes of a DPP row are 16 consecutive rows of ONE 16-byte piece (g = piece + L * (row / 16)),
    // so "the row before" is a row_shr -- no LDS crossbar trip per shuffle
    const int lane = tid & 63;
    const int ri = lane & 15, g = lane >> 4, l = g % L, rl = (tid >> 6) * WROWS + (g / L) * 16 + ri;
    const float *go = grad_out + (size_t)bi * m_ns * c + ch0 + 4 * l;
    const int *ix = idx + (size_t)bi * m_ns;
    auto shr_f = [](float x, auto D) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x110 + decltype(D)::value, 0xf, 0xf, true)); };
    auto load = [&](float4 (&v)[U], int (&ii)[U], int rb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = rb + u * PASS + rl;
            ii[u] = -1;
            if (r < m_ns) { ii[u] = ix[r]; v[u] = *reinterpret_cast<const float4 *>(go + (size_t)r * c); }
        }
    };
    // A padded index list repeats one index over its tail: rows of equal index are added up inside the wavefront first (a segmented scan
    // over the RUNS of equal index among its 16 rows; a later run of the same index adds separately) and one lane per run and piece goes
    // to LDS.  Sixteen lanes adding to one LDS word are sixteen serialised updates otherwise (176 us against 60).
    auto add = [&](const float4 (&v)[U], const int (&ii)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int key = ii[u];
            float4 w = (key >= 0 && key < n) ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (key >= n) key = -1;
            int run = 1;              // rows of my run up to and including mine
            auto step = [&](auto D) {
                constexpr int d = decltype(D)::value;
                const int okey = __builtin_amdgcn_update_dpp(-2, key, 0x110 + d, 0xf, 0xf, false);
                const int orun = __builtin_amdgcn_update_dpp(0, run, 0x110 + d, 0xf, 0xf, false);
                const float ox = shr_f(w.x, D), oy = shr_f(w.y, D), oz = shr_f(w.z, D), ow = shr_f(w.w, D);
                if (ri >= d && okey == key && run == d) {      // my partial is exactly the d rows up to mine, and the one before continues the run
                    w.x += ox; w.y += oy; w.z += oz; w.w += ow;
                    run += orun;
                }
            };
            step(std::integral_constant<int, 1>()); step(std::integral_constant<int, 2>());
            step(std::integral_constant<int, 4>()); step(std::integral_constant<int, 8>());
            const int nkey = __builtin_amdgcn_update_dpp(-2, key, 0x100 + 1, 0xf, 0xf, false);      // row_shl:1 -- the next row's index
            const bool last = ri == 15 || nkey != key;
            if (key >= 0 && last) {
                float *a = gp_acc + key * SLICE + 4 * l;
                atomicAdd(a + 0, w.x); atomicAdd(a + 1, w.y); atomicAdd(a + 2, w.z); atomicAdd(a + 3, w.w);
            }
        }
    };
    {
        float4 va[U], vb[U];
        int ia[U], ib[U];
        constexpr int STEP = U * PASS;
        load(va, ia, 0);
        for (int rb = 0; rb < m_ns; rb += 2 * STEP) {
            load(vb, ib, rb + STEP);              // (rows past the end: index -1, nothing loaded)
            add(va, ia);
            load(va, ia, rb + 2 * STEP);
            add(vb, ib);
        }
    }
    __syncthreads();
    float *gp = grad_points + (size_t)bi * n * c + ch0;
    for (int e = tid; e < n * L; e += 512) {             // (point, 16-byte piece of the slice)
        const int p = e / L, q = e % L;
        float4 sum = *reinterpret_cast<const float4 *>(gp_acc + p * SLICE + 4 * q);
        float4 *dst = reinterpret_cast<float4 *>(gp + (size_t)p * c + 4 * q);
        if (accumulate) { const float4 o = *dst; sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w; }      // (the op ADDS into grad_points, which the caller zeroed)
        *dst = sum;
    }
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


// ---- partial selection sort with the row in REGISTERS (tf_grouping_g.cu:83-123, exact swap sequence) ----
// One wavefront per row: position t lives in register t / 64 of lane t % 64, so a pass is NR compares per lane, one wave
// arg-min and a two-element swap -- no memory traffic inside the k passes (the first version re-read the row from HBM in
// every pass and serialised on a fence).  The original indices ride in a ushort array in LDS, touched by lane 0 only.
//   FUSED = false: selectionSortLauncher semantics -- the row comes from dist[b,m,n]; the whole permuted row and index array
//                  are written back (the reference op's outputs are full-size, only the first k are meaningful);
//   FUSED = true:  knn_point (tf_grouping.py:48-73) in one kernel -- the row is computed from the coordinates as
//                  sum_c (xyz1 - xyz2)^2 (same order as lrg_pairwise_sqdist) and only the first k (value, index) pairs are
//                  written: the b x m x n matrix never exists.
// The reference scans t = s+1 .. n-1 keeping `min` while dist[t] < dist[min] (strict): the FIRST minimum, NaNs never
// preferred, and a NaN at position s stays put.
// Wave-wide minimum through DPP (the row's 16 lanes by quad_perm / mirrors, rows by row_bcast:15 / :31; the result is read from lane 63: uniform, a scalar).
__device__ __forceinline__ float lrg_wave_min_f32(float x) {      // x must not be NaN
    // v_min_f32 with the DPP modifier on its first operand: one instruction per step (through the builtins: a copy, a DPP move, a canonicalising maximum and the
    // minimum); a VGPR written by the previous vector instruction needs two wait states before a DPP read.  Lanes of masked-out rows keep x.
    asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(x));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
// The row as ONE vector value (32 registers at most: a VGPR tuple; 64 registers = two of them): elements are read with constant indices everywhere, and the one
// write whose register is known only at run time -- as a scalar -- compiles to s_set_gpr_idx_on + a move (an array indexed like that goes to scratch memory; a
// block per register that writes "its" element makes the compiler carry the whole row through copies at the 32-way merge: profiles/r04_rowselect_ab.txt).
template <int NR>
struct LrgRow {
    static constexpr int H = NR > 32 ? 32 : NR;
    typedef float V __attribute__((ext_vector_type(H)));
    V lo, hi;      // (hi: registers 32 .. 63 of a 64-register row)
    __device__ __forceinline__ float get(int r) const { return (NR > 32 && r >= 32) ? hi[r >= 32 ? r - 32 : 0] : lo[r < H ? r : 0]; }      // r: a constant after unrolling
    __device__ __forceinline__ void set(int r, float x) {
        if (NR > 32 && r >= 32) hi[r >= 32 ? r - 32 : 0] = x;
        else lo[r < H ? r : 0] = x;
    }
    // lane `mine` of register rp (a scalar) takes val
    __device__ __forceinline__ void put(int rp, bool mine, float val) {
        if (NR > 32 && rp >= 32) { const float o = hi[rp - 32]; hi[rp - 32] = mine ? val : o; }
        else { const float o = lo[rp]; lo[rp] = mine ? val : o; }
    }
};

// The search of lrg_rowselect_passes for the first position >= s holding `best`: R = the first register of the chunk looked at (four compares whose lane masks
// are scalars, one branch per chunk: a branch per register was sixteen dependent vector-to-scalar round trips per pass on average), RS = the register of
// position s.  Scalars only.
template <int R, int RS, int NR>
__device__ __forceinline__ int lrg_rowselect_find(const LrgRow<NR> &v, float best, int ls) {
    if constexpr (R < NR) {
        constexpr int CH = 4;
        unsigned long long m[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            m[j] = (R + j < NR) ? __ballot(v.get(R + j < NR ? R + j : R) == best) : 0ull;
            if (R + j == RS) m[j] &= ~0ull << ls;                   // (positions below s are out)
        }
        if (m[0] | m[1] | m[2] | m[3]) {
            if (m[0]) return R * 64 + __builtin_ctzll(m[0]);
            if (m[1]) return (R + 1) * 64 + __builtin_ctzll(m[1]);
            if (m[2]) return (R + 2) * 64 + __builtin_ctzll(m[2]);
            return (R + 3) * 64 + __builtin_ctzll(m[3]);
        }
        return lrg_rowselect_find<R + CH, RS, NR>(v, best, ls);
    } else {
        return -1;
    }
}

#define LRG_ROWSELECT_MAX_RS 8      // register rows that can hold selected positions: k <= 512 (larger k: the memory-resident kernel)
template <int RS, int NR>
__device__ __forceinline__ void lrg_rowselect_passes(LrgRow<NR> &v, unsigned short *orig, int kk, int L) {
    if constexpr (RS < NR && RS < LRG_ROWSELECT_MAX_RS) {
        constexpr int rs = RS;
        if (rs * 64 >= kk) return;
        for (int ls = 0; ls < 64; ++ls) {
            const int s = rs * 64 + ls;
            if (s >= kk) break;
            const float vrs = v.get(rs);
            const float vs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vrs), ls));      // the value at position s (a scalar)
            // The minimum VALUE first (v_min3 over the registers, a DPP reduction over the lanes; NaNs ignored as by the reference's strict compare) -- a
            // (value, position) pair carried through the scan and the wave reduction was three instructions per register and ten per reduction step.
            float mv = L >= ls ? vrs : INFINITY;
            if (!(mv == mv)) mv = INFINITY;
#pragma unroll
            for (int r = rs + 1; r < NR; ++r) mv = fminf(mv, v.get(r));
            const float best = lrg_wave_min_f32(mv);
            // nothing below +inf (all remaining +inf / NaN), or a NaN at s (nothing compares below it): position s stays
            if (!(best < INFINITY) || vs != vs) continue;
            // ... then the first POSITION that holds it, from ballot masks in register order, and the exchange
            const int p = lrg_rowselect_find<RS, RS, NR>(v, best, ls);
            if (p >= 0 && p != s) {
                v.put(p >> 6, L == (p & 63), vs);
                v.set(rs, L == ls ? best : v.get(rs));
                if (L == 0) { const unsigned short t0 = orig[p]; orig[p] = orig[s]; orig[s] = t0; }
            }
        }
        lrg_rowselect_passes<RS + 1, NR>(v, orig, kk, L);
    }
}

template <int NR, bool FUSED>
__global__ __launch_bounds__(256) void lrg_rowselect_kernel(long rows, int n, int m, int c, int k, const float *dist, const float *xyz1,
                                                             const float *xyz2, int *outi, float *out) {
    __shared__ unsigned short orig[4][NR * 64];
    const int w = threadIdx.x >> 6, L = lrg_lane();
    const long q = (long)blockIdx.x * 4 + w;
    if (q >= rows) return;
    LrgRow<NR> v;
    if (FUSED) {
        const long bi = q / m;
        const float *p1 = xyz1 + bi * n * c, *p2 = xyz2 + q * c;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int t = r * 64 + L;
            float acc = INFINITY;
            if (t < n) {
                for (int l = 0; l < c; ++l) {
                    const float d = __fsub_rn(p1[(long)t * c + l], p2[l]);
                    const float sq = __fmul_rn(d, d);
                    acc = l == 0 ? sq : __fadd_rn(acc, sq);
                }
            }
            v.set(r, acc);
        }
    } else {
        const float *d = dist + q * n;
#pragma unroll
        for (int r = 0; r < NR; ++r) { const int t = r * 64 + L; v.set(r, t < n ? d[t] : INFINITY); }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) orig[w][r * 64 + L] = (unsigned short)(r * 64 + L);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int kk = min(k, n);
    // the passes over the positions 64 rs .. 64 rs + 63, one instantiation per rs (unrolled by the template, not left to the unroller's size limits)
    lrg_rowselect_passes<0, NR>(v, orig[w], kk, L);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (FUSED) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int t = r * 64 + L;
            if (t < kk) { out[q * k + t] = v.get(r); outi[q * k + t] = orig[w][t]; }
        }
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int t = r * 64 + L;
            if (t < n) { out[q * n + t] = v.get(r); outi[q * n + t] = orig[w][t]; }
        }
    }
}

template <bool FUSED>
static int launch_rowselect(long rows, int n, int m, int c, int k, const float *dist, const float *xyz1, const float *xyz2, int *outi,
                            float *out, hipStream_t st) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (k > 64 * LRG_ROWSELECT_MAX_RS) return 1;      // (more selected positions than the register formulation is instantiated for)
    if (n <= 512) hipLaunchKernelGGL((lrg_rowselect_kernel<8, FUSED>), grid, block, 0, st, rows, n, m, c, k, dist, xyz1, xyz2, outi, out);
    else if (n <= 1024) hipLaunchKernelGGL((lrg_rowselect_kernel<16, FUSED>), grid, block, 0, st, rows, n, m, c, k, dist, xyz1, xyz2, outi, out);
    else if (n <= 2048) hipLaunchKernelGGL((lrg_rowselect_kernel<32, FUSED>), grid, block, 0, st, rows, n, m, c, k, dist, xyz1, xyz2, outi, out);
    else if (n <= 4096) hipLaunchKernelGGL((lrg_rowselect_kernel<64, FUSED>), grid, block, 0, st, rows, n, m, c, k, dist, xyz1, xyz2, outi, out);
    else return 1;                      // rows above 4096 entries: the caller takes the memory-resident formulation
    LRG_LAUNCH_CHECK();
    return 0;
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

// ---- ball query + the two gathers of sample_and_group (train_pointnet.py:113-121: query_ball_point, group_point(xyz), group_point(points)) in ONE launch ----
// The ball query at the reference's harness shape is a 5.8 us kernel on a 1.3 MB problem: bound by its launch and one trip to memory, not by bytes -- and the reference
// never calls it alone: the two group_point launches that follow read the index list it has just written.  Here the wavefront that scanned a query keeps the list in LDS
// and gathers both row sets itself: the same idx / pts_cnt, the same gathered values (optionally with the query's coordinates subtracted from the gathered xyz, the
// "translation normalization" of :117, a float32 subtraction).
__global__ __launch_bounds__(256) void lrg_ball_group_kernel(int b, int n, int m, int c, float radius, int nsample, const float *xyz1, const float *xyz2,
                                                             const float *points, int *idx, int *pts_cnt, float *gxyz, float *gpts, int subtract_center) {
    extern __shared__ int lrg_bg_list[];                              // [4 wavefronts][nsample]
    const int wave = threadIdx.x >> 6, lane = lrg_lane();
    const long q = (long)blockIdx.x * 4 + wave;
    if (q >= (long)b * m) return;
    int *list = lrg_bg_list + wave * nsample;
    const long bi = q / m;
    const float *p1 = xyz1 + bi * n * 3;
    const float x2 = xyz2[q * 3 + 0], y2 = xyz2[q * 3 + 1], z2 = xyz2[q * 3 + 2];
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    int cnt = 0, first = -1;
    constexpr int UN = 4;
    for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64 * UN) {      // (lrg_query_ball_kernel's scan: the FIRST nsample points inside the radius, tf_grouping_g.cu:3-36)
        float px[UN], py[UN], pz[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = min(k0 + 64 * u + lane, n - 1);
            px[u] = p1[k * 3 + 0]; py[u] = p1[k * 3 + 1]; pz[u] = p1[k * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = k0 + 64 * u + lane;
            if (k0 + 64 * u >= n || cnt >= nsample) break;
            bool hit = false;
            if (k < n) {
                float dx = __fsub_rn(x2, px[u]), dy = __fsub_rn(y2, py[u]), dz = __fsub_rn(z2, pz[u]);
                float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                float d = fmaxf(__fsqrt_rn(d2), 1e-20f);
                hit = d < radius;
            }
            unsigned long long mask = __ballot(hit);
            if (mask) {
                if (first < 0) first = k0 + 64 * u + (int)__ffsll((long long)mask) - 1;
                int pos = cnt + __popcll(mask & lt);
                if (hit && pos < nsample) list[pos] = k;
                cnt += __popcll(mask);
            }
        }
    }
    if (cnt > nsample) cnt = nsample;
    for (int l = cnt + lane; l < nsample; l += 64) list[l] = first < 0 ? 0 : first;      // (:26-29)
    if (lane == 0) pts_cnt[q] = cnt;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int *out = idx + q * nsample;
    for (int l = lane; l < nsample; l += 64) out[l] = list[l];
    // group_point(xyz, idx) (:116) -- three floats per sample, consecutive lanes on consecutive floats
    float *ox = gxyz + q * nsample * 3;
    const float cen[3] = {subtract_center ? x2 : 0.f, subtract_center ? y2 : 0.f, subtract_center ? z2 : 0.f};
    for (int e = lane; e < nsample * 3; e += 64) {
        const int l = e / 3, d = e - 3 * l;
        const float v = p1[(long)list[l] * 3 + d];
        ox[e] = subtract_center ? __fsub_rn(v, cen[d]) : v;
    }
    // group_point(points, idx) (:119)
    if (points && gpts) {
        const float *pp = points + bi * n * c;
        float *op = gpts + q * nsample * c;
        if ((c & 3) == 0 && ((((uintptr_t)points | (uintptr_t)gpts) & 15) == 0)) {
            const int c4 = c >> 2;
            for (int e = lane; e < nsample * c4; e += 64) {
                const int l = e / c4, k = e - l * c4;
                reinterpret_cast<float4 *>(op)[e] = reinterpret_cast<const float4 *>(pp + (long)list[l] * c)[k];
            }
        } else {
            for (int e = lane; e < nsample * c; e += 64) {
                const int l = e / c, k = e - l * c;
                op[e] = pp[(long)list[l] * c + k];
            }
        }
    }
}

int lrg_query_ball_group(int b, int n, int m, int c, float radius, int nsample, const float *xyz1, const float *xyz2, const float *points, int *idx,
                         int *pts_cnt, float *grouped_xyz, float *grouped_points, int subtract_center, void *stream) {
    if (b < 0 || n <= 0 || m < 0 || nsample <= 0 || c < 0 || !xyz1 || !xyz2 || !idx || !pts_cnt || !grouped_xyz) return LRG_EINVAL - 1;
    if ((points != nullptr) != (grouped_points != nullptr) || (points && c <= 0) || nsample > 4096) return LRG_EINVAL - 2;
    long q = (long)b * m;
    if (q == 0) return 0;
    hipLaunchKernelGGL(lrg_ball_group_kernel, dim3((unsigned)((q + 3) / 4)), dim3(256), (size_t)4 * nsample * sizeof(int), (hipStream_t)stream, b, n, m, c, radius,
                       nsample, xyz1, xyz2, points, idx, pts_cnt, grouped_xyz, grouped_points, subtract_center);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream) {
    if (b < 0 || n <= 0 || m < 0 || k < 0 || !dist || !outi || !out) return LRG_EINVAL - 1;
    long rows = (long)b * m;
    if (rows == 0) return 0;
    const int rc = launch_rowselect<false>(rows, n, m, 0, k, dist, nullptr, nullptr, outi, out, (hipStream_t)stream);
    if (rc <= 0) return rc;
    hipLaunchKernelGGL(lrg_selection_sort_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       rows, n, k, dist, outi, out);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_knn_topk(int b, int n, int m, int c, int k, const float *xyz1, const float *xyz2, float *val, int *idx, void *stream) {
    if (b < 0 || n <= 0 || m < 0 || c <= 0 || k <= 0 || k > n || !xyz1 || !xyz2 || !val || !idx) return LRG_EINVAL - 1;
    if (n > 4096 || k > 64 * LRG_ROWSELECT_MAX_RS) return LRG_EINVAL - 2;          // larger rows or more neighbours: lrg_pairwise_sqdist + lrg_selection_sort
    long rows = (long)b * m;
    if (rows == 0) return 0;
    const int rc = launch_rowselect<true>(rows, n, m, c, k, nullptr, xyz1, xyz2, idx, val, (hipStream_t)stream);
    return rc > 0 ? LRG_EINVAL - 2 : rc;
}

int lrg_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out,
                    void *stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || !points || !idx || !out) return LRG_EINVAL - 1;
    long total = (long)b * m * nsample * c;
    if (total == 0) return 0;
    if (c % 4 == 0 && total / 4 < 0x7fffffffL && (((uintptr_t)points | (uintptr_t)out) & 15) == 0 && (long)m * nsample < 0x7fffffffL) {
        const unsigned total4 = (unsigned)(total / 4);
        hipLaunchKernelGGL(lrg_group_point_vec4_kernel, dim3((total4 + 256 * LRG_GP_U - 1) / (256 * LRG_GP_U)), dim3(256), 0, (hipStream_t)stream, total4, n, c / 4,
                           m * nsample, reinterpret_cast<const float4 *>(points), idx, reinterpret_cast<float4 *>(out));
        LRG_LAUNCH_CHECK();
        return 0;
    }
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
    const size_t lds = (size_t)n * LRG_GPG_SLICE * sizeof(float);
    const long m_ns = (long)m * nsample;
    if (c % LRG_GPG_SLICE == 0 && c / LRG_GPG_SLICE <= 4096 && lds <= 144 * 1024 && m_ns < (1L << 30) && b <= 65535 &&
        ((((uintptr_t)grad_out) | ((uintptr_t)grad_points)) & 15) == 0 && m_ns >= 256) {
        auto kern = lrg_group_point_grad_lds_kernel<LRG_GPG_SLICE, LRG_GPG_U>;
        static bool attr_done[LRG_MAX_DEVICES] = {};
        const int dev = lrg_current_device();
        if (!attr_done[dev]) {
            LRG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            attr_done[dev] = true;
        }
        const int nslice = c / LRG_GPG_SLICE;
        hipLaunchKernelGGL(kern, dim3((unsigned)(((b + 7) / 8) * 8 * nslice)), dim3(512), lds, (hipStream_t)stream, b, n, c, (int)m_ns, grad_out, idx, grad_points, 1);
        LRG_LAUNCH_CHECK();
        return 0;
    }
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
