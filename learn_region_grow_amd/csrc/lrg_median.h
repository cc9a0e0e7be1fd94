// Per-channel median of a region's points (numpy.median, test_region_grow.py:241): selection on order-preserving keys.
// Shared by the loop kernels (lrg_grow.hip, lrg_front.inl) and by the median workgroups that ride in the packed branch launch
// (lrg_fused.hip).
#pragma once
#include "lrg_common.h"

#ifndef LRG_MED48_BISECT
#define LRG_MED48_BISECT 2      // regions above 16 Ki points (48 register keys per thread): 2 = two sampled pivots, then bisection among
                                // the keys between them; 1 = bisection over all keys; 0 = radix select
#endif

__device__ __forceinline__ uint32_t lrg_f2key(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float lrg_key2f(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}


// k-th smallest (0-based) keys for two ranks at once, by bitwise bisection on the key: the answer is the largest r
// with #(key < r) <= k.  32 counting passes, no atomics, no sorting (regions of 1..10^4 points, test_region_grow.py:241).
__device__ void lrg_select2(const uint32_t *cache, bool cached, const float *pts, const int32_t *idx, int F, int ch,
                            int nc, int ka, int kb, int *sh, uint32_t *ra_out, uint32_t *rb_out) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nround = (nc + blockDim.x - 1) / blockDim.x;
    // The keys of one channel of one region are clustered (coordinates within a room, near-constant normals): their
    // common high bits cannot discriminate, so find them first (one min/max pass) and bisect only the bits below.
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    for (int it = 0; it < nround; ++it) {
        int j = it * blockDim.x + threadIdx.x;
        if (j < nc) {
            uint32_t key = cached ? cache[j] : lrg_f2key(pts[(long)idx[j] * F + ch]);
            kmin = min(kmin, key); kmax = max(kmax, key);
        }
    }
    kmin = lrg_wave_min_u32(kmin);
    kmax = lrg_wave_max_u32(kmax);
    if (lrg_lane() == 0) { sh[2 * wave] = (int)kmin; sh[2 * wave + 1] = (int)kmax; }
    __syncthreads();
    for (int w = 0; w < nw; ++w) { kmin = min(kmin, (uint32_t)sh[2 * w]); kmax = max(kmax, (uint32_t)sh[2 * w + 1]); }
    __syncthreads();
    const uint32_t diff = kmin ^ kmax;
    const int hb = diff ? 32 - __clz((int)diff) : 0;              // bits [hb,32) are common to every key
    const uint32_t common = hb >= 32 ? 0u : (kmin >> hb) << hb;
    uint32_t ra = common, rb = common;
    for (int bit = hb - 1; bit >= 0; --bit) {
        const uint32_t ca = ra | (1u << bit), cb = rb | (1u << bit);
        int cnt_a = 0, cnt_b = 0;
        for (int it = 0; it < nround; ++it) {
            int j = it * blockDim.x + threadIdx.x;
            if (j < nc) {
                uint32_t key = cached ? cache[j] : lrg_f2key(pts[(long)idx[j] * F + ch]);
                cnt_a += key < ca ? 1 : 0;
                cnt_b += key < cb ? 1 : 0;
            }
        }
        cnt_a = lrg_wave_sum_i32(cnt_a);
        cnt_b = lrg_wave_sum_i32(cnt_b);
        if (lrg_lane() == 0) { sh[2 * wave] = cnt_a; sh[2 * wave + 1] = cnt_b; }
        __syncthreads();
        int ta = 0, tb = 0;
        for (int w = 0; w < nw; ++w) { ta += sh[2 * w]; tb += sh[2 * w + 1]; }
        __syncthreads();
        if (ta <= ka) ra = ca;
        if (tb <= kb) rb = cb;
    }
    *ra_out = ra; *rb_out = rb;
}


// Bisection select of the two middle ranks over R register keys per lane (padding keys = 0xFFFFFFFF never count):
// per-lane VALU counters and one DPP wave reduction per step -- no LDS, no barriers, no scalar popcounts.
template <int R>
__device__ __forceinline__ float lrg_select_regs(const uint32_t (&key)[R], int nc) {
    const int lane = lrg_lane();
    // common high bits of the (clustered) keys cannot discriminate: bisect only below them
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r * 64 + lane < nc) { kmin = min(kmin, key[r]); kmax = max(kmax, key[r]); }
    kmin = lrg_wave_min_u32(kmin);
    kmax = lrg_wave_max_u32(kmax);
    const uint32_t diff = kmin ^ kmax;
    const int hb = diff ? 32 - __clz((int)diff) : 0;
    const uint32_t common = hb >= 32 ? 0u : (kmin >> hb) << hb;
    // upper median = element of rank k2 = the largest r with #(key < r) <= k2; one compare per key per step
    const int k2 = nc >> 1;
    uint32_t rb = common;
    for (int bit = hb - 1; bit >= 0; --bit) {
        const uint32_t cb = rb | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) cnt += key[r] < cb ? 1 : 0;        // padding keys (0xFFFFFFFF) never count
        if (lrg_wave_sum_i32(cnt) <= k2) rb = cb;
    }
    float hi = lrg_key2f(rb);
    if (nc & 1) return hi;
    // even count: the element of rank k2-1 is rb itself when fewer than k2 keys lie below rb (duplicates of rb span
    // both ranks), otherwise it is the largest key below rb
    int below = 0;
    uint32_t mx = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (key[r] < rb) { ++below; mx = max(mx, key[r]); }
    below = lrg_wave_sum_i32(below);
    mx = lrg_wave_max_u32(mx);
    float lo = below >= k2 ? lrg_key2f(mx) : hi;
    return __fmul_rn(__fadd_rn(lo, hi), 0.5f);                          // numpy.mean of the two middle float32 values
}

// Median of one channel over nc <= 64*R current points by ONE wavefront, keys gathered straight from HBM.
// All loads are unconditional at clamped positions: predicated loads would each sit in their own branch with a full
// s_waitcnt behind it (R dependent round trips instead of 2).
template <int R>
__device__ __forceinline__ float lrg_median_wave_r(const float *pts, const int32_t *idx, int F, int nc) {
    const int lane = lrg_lane();
    uint32_t key[R];
    {
        int id[R];
#pragma unroll
        for (int r = 0; r < R; ++r) id[r] = idx[min(r * 64 + lane, nc - 1)];
#pragma unroll
        for (int r = 0; r < R; ++r) key[r] = (r * 64 + lane < nc) ? lrg_f2key(pts[(long)id[r] * F]) : 0xFFFFFFFFu;
    }
    return lrg_select_regs<R>(key, nc);
}


// Selection over KT register keys per thread of a BT-thread workgroup (key[r] = the key of position r * BT + tid, positions >= n
// hold 0xFFFFFFFF): the key of rank k2 and, if need_lo, its mean with the key of rank k2 - 1 (numpy.median of an even count).
// Bisection with per-thread counters, a DPP wave sum and one LDS atomic + one barrier per step.
// sh[0] = min (caller: 0xFFFFFFFF), sh[1] = max, sh[2..49] = three counters per step, sh[50] = below-count, sh[51] = max below
// (caller: 0), visible to the workgroup (a barrier after the initialisation).
template <int KT, int BT>
__device__ __forceinline__ float lrg_block_select_regs(const uint32_t (&key)[KT], int n, int k2, bool need_lo, int *sh) {
    const int tid = threadIdx.x, lane = lrg_lane();
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int r = 0; r < KT; ++r)
        if (r * BT + tid < n) { kmin = min(kmin, key[r]); kmax = max(kmax, key[r]); }
    kmin = lrg_wave_min_u32(kmin);
    kmax = lrg_wave_max_u32(kmax);
    if (lane == 0) { atomicMin(reinterpret_cast<unsigned *>(&sh[0]), kmin); atomicMax(reinterpret_cast<unsigned *>(&sh[1]), kmax); }
    __syncthreads();
    kmin = (uint32_t)sh[0]; kmax = (uint32_t)sh[1];
    const uint32_t diff = kmin ^ kmax;
    const int hb = diff ? 32 - __clz((int)diff) : 0;
    const uint32_t common = hb >= 32 ? 0u : (kmin >> hb) << hb;
    uint32_t rb = common;
    // two bits per step on aligned bit pairs (three thresholds, three counters of up to 48 Ki each).
    // A pair that reaches into the common prefix needs no special case: a threshold that would flip a common bit counts
    // either every key or the same keys as a lower threshold.
    int slot = 2;
    for (int bit = ((hb + 1) & ~1) - 2; bit >= 0; bit -= 2, slot += 3) {
        const uint32_t t1 = rb | (1u << bit), t2 = rb | (2u << bit), t3 = rb | (3u << bit);
        int c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
        for (int r = 0; r < KT; ++r) {
            c1 += key[r] < t1 ? 1 : 0;
            c2 += key[r] < t2 ? 1 : 0;
            c3 += key[r] < t3 ? 1 : 0;
        }
        c1 = lrg_wave_sum_i32(c1);
        c2 = lrg_wave_sum_i32(c2);
        c3 = lrg_wave_sum_i32(c3);
        if (lane == 0) { if (c1) atomicAdd(&sh[slot], c1); if (c2) atomicAdd(&sh[slot + 1], c2); if (c3) atomicAdd(&sh[slot + 2], c3); }
        __syncthreads();
        const int s1 = sh[slot], s2 = sh[slot + 1], s3 = sh[slot + 2];
        if (s3 <= k2) rb = t3;
        else if (s2 <= k2) rb = t2;
        else if (s1 <= k2) rb = t1;
    }
    float hi = lrg_key2f(rb);
    if (!need_lo) return hi;
    // the key of rank k2 - 1 is rb itself when fewer than k2 keys lie below rb (duplicates of rb span both ranks), else the largest below
    int below = 0;
    uint32_t mx = 0u;
#pragma unroll
    for (int r = 0; r < KT; ++r)
        if (key[r] < rb) { ++below; mx = max(mx, key[r]); }
    below = lrg_wave_sum_i32(below);
    mx = lrg_wave_max_u32(mx);
    if (lane == 0) { if (below) atomicAdd(&sh[50], below); atomicMax(reinterpret_cast<unsigned *>(&sh[51]), mx); }
    __syncthreads();
    float lo = sh[50] >= k2 ? lrg_key2f((uint32_t)sh[51]) : hi;
    return __fmul_rn(__fadd_rn(lo, hi), 0.5f);
}

// Larger regions: one BT-thread workgroup per (slot, channel), KT keys per thread in REGISTERS (two global round trips in all).
template <int KT, int BT = 1024>
__device__ __forceinline__ float lrg_median_block_regs(const float *pts, const int32_t *idx, int F, int nc, int *sh) {
    const int tid = threadIdx.x;
    uint32_t key[KT];
    {
        int id[KT];
#pragma unroll
        for (int r = 0; r < KT; ++r) id[r] = idx[min(r * BT + tid, nc - 1)];
#pragma unroll
        for (int r = 0; r < KT; ++r) key[r] = (r * BT + tid < nc) ? lrg_f2key(pts[(long)id[r] * F]) : 0xFFFFFFFFu;
    }
    return lrg_block_select_regs<KT, BT>(key, nc, nc >> 1, !(nc & 1), sh);
}

// The key of rank k among n <= 64 * R keys held R per lane by ONE wavefront (padding 0xFFFFFFFF): bisection, no LDS.
template <int R>
__device__ __forceinline__ uint32_t lrg_select_rank_wave(const uint32_t (&key)[R], int n, int k) {
    const int lane = lrg_lane();
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r * 64 + lane < n) { kmin = min(kmin, key[r]); kmax = max(kmax, key[r]); }
    kmin = lrg_wave_min_u32(kmin);
    kmax = lrg_wave_max_u32(kmax);
    const uint32_t diff = kmin ^ kmax;
    const int hb = diff ? 32 - __clz((int)diff) : 0;
    uint32_t rb = hb >= 32 ? 0u : (kmin >> hb) << hb;
    for (int bit = hb - 1; bit >= 0; --bit) {
        const uint32_t cb = rb | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) cnt += key[r] < cb ? 1 : 0;
        if (lrg_wave_sum_i32(cnt) <= k) rb = cb;
    }
    return rb;
}

// Regions of tens of thousands of points (KT = 48 keys per thread): a bisection step over all of them costs 48 x 3 compares per
// thread -- ~5 k cycles for the 16 wavefronts of a CU -- and there are 10-16 of them (a 40 k-point ground region of a KITTI scene:
// 53 us per channel).  Two pivots bracket the median first: the keys of ranks BT/2 -+ LRG_MED_SAMPLE_D of a systematic sample of
// BT keys (one per thread, evenly spaced over the list; each rank found by one wavefront, 16 keys per lane), ONE counting pass
// over all keys (how many lie below the bracket, how many inside), and the keys inside -- an eighth of them -- go to LDS, where
// the ranks wanted are selected among KT2 <= 8 keys per thread.  Exact: when the bracket misses the ranks (the sample rank of the
// true median is binomial, sigma = 16: beyond 4 sigma) or overflows its buffer, the full bisection runs.
// sh: 64 ints as for lrg_block_select_regs, then BT sample keys, then CAP bracket keys.
#define LRG_MED_SAMPLE_D 64
#define LRG_MED_BRACKET_CAP 8192
#define LRG_SAMPLED_LDS_INTS(BT) (64 + (BT) + LRG_MED_BRACKET_CAP)
template <int KT, int BT = 1024, int CAP = LRG_MED_BRACKET_CAP>
__device__ __forceinline__ float lrg_median_block_sampled(const float *pts, const int32_t *idx, int F, int nc, int *sh) {
    static_assert(BT == 1024, "the sample is selected 16 keys per lane");
    static_assert(CAP % BT == 0, "whole keys per thread in the second stage");
    const int tid = threadIdx.x, lane = lrg_lane(), wave = tid >> 6;
    uint32_t *samp = reinterpret_cast<uint32_t *>(sh + 64), *brk = samp + BT;
    uint32_t key[KT];
    uint32_t skey;
    {
        int id[KT];
        const int sid = idx[(int)(((long)tid * nc) / BT)];
#pragma unroll
        for (int r = 0; r < KT; ++r) id[r] = idx[min(r * BT + tid, nc - 1)];
        skey = lrg_f2key(pts[(long)sid * F]);
#pragma unroll
        for (int r = 0; r < KT; ++r) key[r] = (r * BT + tid < nc) ? lrg_f2key(pts[(long)id[r] * F]) : 0xFFFFFFFFu;
    }
    samp[tid] = skey;
    if (tid < 64) sh[tid] = tid == 0 ? -1 : 0;
    __syncthreads();
    if (wave < 2) {                                              // the two pivots, one wavefront each
        uint32_t sk[BT / 64];
#pragma unroll
        for (int r = 0; r < BT / 64; ++r) sk[r] = samp[r * 64 + lane];
        const uint32_t piv = lrg_select_rank_wave<BT / 64>(sk, BT, wave == 0 ? BT / 2 - LRG_MED_SAMPLE_D : BT / 2 + LRG_MED_SAMPLE_D);
        if (lane == 0) sh[60 + wave] = (int)piv;
    }
    __syncthreads();
    const uint32_t plo = (uint32_t)sh[60], phi = (uint32_t)sh[61];
    int below = 0, inside = 0;
#pragma unroll
    for (int r = 0; r < KT; ++r) {
        below += key[r] < plo ? 1 : 0;                           // (padding keys are above every pivot)
        inside += (key[r] >= plo && key[r] <= phi && r * BT + tid < nc) ? 1 : 0;
    }
    // workgroup totals, and this thread's place in the bracket buffer (exclusive scan of `inside`: wavefront scan + wavefront offsets)
    const int incl = lrg_wave_incl_scan_i32(inside);
    const int wbelow = lrg_wave_sum_i32(below);
    if (lane == 63) sh[32 + wave] = incl;
    if (lane == 0 && wbelow) atomicAdd(&sh[62], wbelow);
    __syncthreads();
    int woff = 0, M = 0;
#pragma unroll
    for (int w = 0; w < BT / 64; ++w) { if (w < wave) woff += sh[32 + w]; M += sh[32 + w]; }
    const int B = sh[62];
    const int k2 = nc >> 1, k1 = (nc & 1) ? k2 : k2 - 1;
    __syncthreads();                                             // (sh[32..] and sh[60..62] are read: the selects below start from a clean sh[0..63])
    if (tid < 64) sh[tid] = tid == 0 ? -1 : 0;
    __syncthreads();
    if (!(B <= k1 && k2 < B + M && M <= CAP))                    // (workgroup-uniform)
        return lrg_block_select_regs<KT, BT>(key, nc, k2, !(nc & 1), sh);
    int pos = woff + incl - inside;
#pragma unroll
    for (int r = 0; r < KT; ++r)
        if (key[r] >= plo && key[r] <= phi && r * BT + tid < nc) brk[pos++] = key[r];
    __syncthreads();
    constexpr int KT2 = CAP / BT;
    uint32_t k2v[KT2];
#pragma unroll
    for (int q = 0; q < KT2; ++q) k2v[q] = q * BT + tid < M ? brk[q * BT + tid] : 0xFFFFFFFFu;
    return lrg_block_select_regs<KT2, BT>(k2v, M, k2 - B, !(nc & 1), sh);
}


// centred channel of grid row y: 0, 1, 6, 7, ... (:243-247); -1 past the feature count
__device__ __forceinline__ int lrg_centred_channel(int y, int F) { const int ch = y < 2 ? y : y + 4; return ch < F ? ch : -1; }

__device__ __forceinline__ bool lrg_is_centred(int ch, int F) { return (ch < 2 || ch >= 6) && ch < F; }

// Where the keys of centred channel ch = lrg_centred_channel(y, F) of a room come from: value of point i = base[i * stride].
// With the room's channel-major copy (LrgRoom::chan_major) a region -- mostly runs of consecutive indices, a room's points come
// object after object -- reads a few dense lines per channel; from the [n,F] rows every key is a line of its own
// (a 40 k-point region: 160 KB against 5 MB per channel).
struct LrgChanSrc { const float *base; int stride; };
__device__ __forceinline__ LrgChanSrc lrg_chan_src(const LrgRoom *R, int y, int ch, int F) {
    LrgChanSrc c;
    const float *cm = R->chan_major;
    if (cm) { c.base = cm + (long)y * R->chan_stride; c.stride = 1; }
    else { c.base = R->points + ch; c.stride = F; }
    return c;
}


// ------------------------------------------------------------------------------------------------------------------------
// Block medians by radix select: NCH channels of one region at once, BT threads, KT keys per thread and channel in registers.
// After the common high bits (min / max pass), 8 bits per pass from the top of the VARYING bits -- the first histogram already
// spreads a region's keys over up to 256 bins and leaves ~n / 256 candidates, so 2-4 passes of (LDS atomics, barrier, one
// wavefront's scan of 256 bins, barrier) replace the 10-16 barrier-separated bisection steps: what a step costs is the barrier
// and the reductions, not the compares.  Same result as lrg_median_block_regs: the key K with #(key < K) <= nc/2 < #(key <= K),
// averaged with the largest key below it (or with itself, duplicates) when nc is even.
// sh: LRG_RADIX_LDS_INTS(NCH) ints of LDS.  out[c] valid in every thread.  Channels ch[c] < 0 are skipped.
// ------------------------------------------------------------------------------------------------------------------------
#define LRG_RADIX_HDR(NCH) ((NCH) <= 3 ? 64 : 256)
#define LRG_RADIX_LDS_INTS(NCH) (LRG_RADIX_HDR(NCH) + (NCH) * 1024)
template <int KT, int BT, int NCH>
__device__ __forceinline__ void lrg_median_block_radix(const float *points, const int (&ch)[NCH], const int32_t *idx, int F, int nc,
                                                       int *sh, float (&out)[NCH]) {
    const int tid = threadIdx.x, lane = lrg_lane(), wave = tid >> 6;
    uint32_t key[NCH][KT];
    {
        int id[KT];
#pragma unroll
        for (int r = 0; r < KT; ++r) id[r] = idx[min(r * BT + tid, nc - 1)];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int r = 0; r < KT; ++r)
                key[c][r] = (ch[c] >= 0 && r * BT + tid < nc) ? lrg_f2key(points[(long)id[r] * F + ch[c]]) : 0xFFFFFFFFu;
    }
    // header: [0, 2 NCH) min / max per channel, [BIN0 ...) bin and rank of every (channel, pass), [LOW0 ...) count and maximum below
    constexpr int HDR = LRG_RADIX_HDR(NCH), BIN0 = NCH <= 3 ? 16 : 32, LOW0 = NCH <= 3 ? 48 : 128;
    static_assert(2 * NCH <= BIN0 && BIN0 + 8 * NCH <= LOW0 && LOW0 + 2 * NCH <= HDR, "header layout");
    for (int i = HDR + tid; i < LRG_RADIX_LDS_INTS(NCH); i += BT) sh[i] = 0;
    if (tid < HDR) sh[tid] = (tid < 2 * NCH && !(tid & 1)) ? -1 : 0;
    __syncthreads();
    static_assert(KT <= 64, "alive masks are 64 bits");
    unsigned long long alive[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        alive[c] = 0;
#pragma unroll
        for (int r = 0; r < KT; ++r) alive[c] |= (r * BT + tid < nc) ? (1ull << r) : 0ull;
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
        for (int r = 0; r < KT; ++r)
            if (r * BT + tid < nc) { kmin = min(kmin, key[c][r]); kmax = max(kmax, key[c][r]); }
        kmin = lrg_wave_min_u32(kmin);
        kmax = lrg_wave_max_u32(kmax);
        if (lane == 0) { atomicMin(reinterpret_cast<unsigned *>(&sh[2 * c]), kmin); atomicMax(reinterpret_cast<unsigned *>(&sh[2 * c + 1]), kmax); }
    }
    __syncthreads();
    uint32_t prefix[NCH];
    int kk[NCH], top[NCH];                                         // rank among the candidates; bits [0, top) still unresolved
    int maxtop = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t kmin = (uint32_t)sh[2 * c], kmax = (uint32_t)sh[2 * c + 1];
        const uint32_t diff = kmin ^ kmax;
        const int hb = diff ? 32 - __clz((int)diff) : 0;
        prefix[c] = hb >= 32 ? 0u : (kmin >> hb) << hb;
        kk[c] = nc >> 1;
        top[c] = ch[c] >= 0 ? hb : 0;
        maxtop = max(maxtop, top[c]);
    }
    for (int pass = 0; pass < 4 && pass * 8 < maxtop; ++pass) {    // (workgroup-uniform)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (top[c] <= 0) continue;
            const int shift = max(top[c] - 8, 0);
            const uint32_t mask = (1u << (top[c] - shift)) - 1u;
            int *hist = sh + HDR + (c * 4 + pass) * 256;
#pragma unroll
            for (int r = 0; r < KT; ++r)
                if (alive[c] >> r & 1ull) atomicAdd(&hist[(key[c][r] >> shift) & mask], 1);
        }
        __syncthreads();
        if (wave < NCH) {                                          // wavefront c finds channel c's bin: 4 bins per lane
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c != wave || top[c] <= 0) continue;
                const int *hist = sh + HDR + (c * 4 + pass) * 256;
                const int4 h = *reinterpret_cast<const int4 *>(hist + 4 * lane);
                const int s4 = h.x + h.y + h.z + h.w;
                const int incl = lrg_wave_incl_scan_i32(s4), excl = incl - s4;
                if (excl <= kk[c] && kk[c] < incl) {
                    int b = 0, before = excl;
                    if (kk[c] >= before + h.x) { before += h.x; b = 1;
                        if (kk[c] >= before + h.y) { before += h.y; b = 2;
                            if (kk[c] >= before + h.z) { before += h.z; b = 3; } } }
                    sh[BIN0 + (c * 4 + pass) * 2] = 4 * lane + b;
                    sh[BIN0 + (c * 4 + pass) * 2 + 1] = before;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (top[c] <= 0) continue;
            const int shift = max(top[c] - 8, 0);
            const uint32_t mask = (1u << (top[c] - shift)) - 1u;
            const int bin = sh[BIN0 + (c * 4 + pass) * 2], before = sh[BIN0 + (c * 4 + pass) * 2 + 1];
            prefix[c] |= (uint32_t)bin << shift;
            kk[c] -= before;
#pragma unroll
            for (int r = 0; r < KT; ++r)
                if (((key[c][r] >> shift) & mask) != (uint32_t)bin) alive[c] &= ~(1ull << r);
            top[c] = shift;
        }
    }
    // even count: the element of rank nc/2 - 1 is the key itself when fewer than nc/2 keys lie below it, else the largest below
    if (!(nc & 1)) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            int below = 0;
            uint32_t mx = 0u;
#pragma unroll
            for (int r = 0; r < KT; ++r)
                if (r * BT + tid < nc && key[c][r] < prefix[c]) { ++below; mx = max(mx, key[c][r]); }
            below = lrg_wave_sum_i32(below);
            mx = lrg_wave_max_u32(mx);
            if (lane == 0) { if (below) atomicAdd(&sh[LOW0 + 2 * c], below); atomicMax(reinterpret_cast<unsigned *>(&sh[LOW0 + 2 * c + 1]), mx); }
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float hi = lrg_key2f(prefix[c]);
        if (nc & 1) out[c] = hi;
        else {
            const float lo = sh[LOW0 + 2 * c] >= (nc >> 1) ? lrg_key2f((uint32_t)sh[LOW0 + 2 * c + 1]) : hi;
            out[c] = __fmul_rn(__fadd_rn(lo, hi), 0.5f);
        }
    }
}
