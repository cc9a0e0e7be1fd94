// One row tile through a fused LrgNet stack (a whole branch, learn_region_grow_util.py:106-123, or a whole head, :138-162):
// the body of lrg_fused_stack_kernel (lrg_fused.hip) as a device function, so that the same code -- the same MFMA sequence,
// the same epilogue arithmetic, bit for bit -- also runs as a task of the free-running region-grow kernel (lrg_async.inl),
// where four wavefronts of a 1024-thread workgroup form a team of their own.
//
// TEAM: who the 256 threads of the tile are and how they meet --
//   LrgWgTeam    the whole workgroup (threadIdx.x, __syncthreads())
//   LrgLdsTeam   wavefronts 4t .. 4t+3 of a larger workgroup, meeting at a counter in LDS (no s_barrier: the other teams of the
//                workgroup are somewhere else in their own tiles)
// COH: the tile's inputs were written, and its outputs will be read, by OTHER workgroups of the SAME launch.  Per-XCD L2s are not
// coherent with each other and a CU's L1 is never refreshed by another CU's stores, so every such word goes write-through /
// is read past the L1: relaxed agent-scope atomics (global_load / global_store ... sc1) on both sides, no fences
// (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 loads may replace the acquire when the producer stored sc1).
#pragma once
#include "lrg_common.h"
#include "lrg_fused.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef LRG_TILE_N16
#define LRG_TILE_N16 0   // 1: the 64 -> 64 layers of single-instance 32-row tiles on v_mfma_f32_16x16x4_f32, all four waves (0: 32 x 32 strips, two waves).
#endif                   // Bit-identical (tests/test_gpu_free_run.py pass with it, same label checksum) and SLOWER: 804-815 k against 871-873 k instance-steps/s in three
                         // variants of the operand path (lane exchange by v_permlane32_swap, two ds_read_b128 per k-group, all weights of the pass requested up front):
                         // the layer's own cycle stamps barely move (5.7 -> 5.3 k, 5.1 -> 4.9 k) and the pooled layer behind it gets 5 % slower
                         // (profiles/r04_tile_n16.txt).  Off; kept as the measured alternative the round-3 review asked for.
#define FBN 128      // output columns per pass: 4 waves side by side, each a (32*RT)x32 strip (RT 32x32 MFMA tiles sharing B)
#define FTHREADS 256 // one wave per SIMD per tile; 2-3 tiles per CU interleave without sharing barriers

struct LrgWgTeam {
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
};

// Four wavefronts of a larger workgroup.  `cnt` is the team's own LDS word (zero before first use); arrivals are counted
// monotonically, a wavefront leaves when the count reaches four times the number of barriers it has been through.
struct LrgLdsTeam {
    int *cnt;
    mutable int target;
    int base;            // first thread of the team within the workgroup
    int *gave_up;        // nullable: global word that is set when a meeting was given up (after ~5 s without the others)
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x - base; }
    __device__ __forceinline__ void sync() const {
        target += 4;
        // this wavefront's LDS traffic is done before it arrives.  NOT its global traffic: inside a tile nothing goes from wavefront to
        // wavefront through global memory, and the weight ring's loads for the NEXT pass are meant to stay in flight across the layer
        // boundary (__syncthreads() waits for vmcnt(0) too and drains them); who needs stores to be out drains explicitly.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0) {
            __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // (bounded: a team that lost a wavefront shows as wrong results and a raised abort word, not as a hung GPU)
            long long t_long = 0;
            for (unsigned spin = 1; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target; ++spin) {
                if ((spin & 4095u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (!t_long) t_long = now;
                    else if (now - t_long > 500000000LL) {      // 100 MHz: five seconds
                        if (gave_up) __hip_atomic_store(gave_up, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
#ifdef LRG_TEAM_SPIN_SLEEP      // (a tight spin: 856 -> 859 k instance-steps/s at 68 rooms in flight; the waiting lane is alone in its wavefront's slot)
                __builtin_amdgcn_s_sleep(1);
#endif
            }
        }
        asm volatile("" ::: "memory");
    }
};

// words handed from workgroup to workgroup inside a launch (COH)
__device__ __forceinline__ float lrg_ld_coh(const float *p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned *>(const_cast<float *>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int lrg_ld_coh(const int *p) {
    return __hip_atomic_load(const_cast<int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 lrg_ld_coh2(const float *p) {      // 8-byte aligned
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<unsigned long long *>(const_cast<float *>(p)), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}
__device__ __forceinline__ void lrg_st_coh(float *p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lrg_st_coh(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16 bytes at once (one fabric transaction instead of two or four: a narrow sc1 store is a write of its own, 452 KB of write traffic
// for an 8 KB tile of conv[1] rows as 8-byte stores, profiles/r03_pmc_free_run.json): through a buffer descriptor on `base` (wave-uniform)
typedef unsigned int lrg_u32x4v __attribute__((ext_vector_type(4)));
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void lrg_st_coh4(float *base, unsigned byte_off, float4 v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    const lrg_u32x4v u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, 16);       // aux 16 = sc1
}
__device__ __forceinline__ float4 lrg_ld_coh4(const float *base, unsigned byte_off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
    const lrg_u32x4v u = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
#else
__device__ __forceinline__ void lrg_st_coh4(float *, unsigned, float4) {}
__device__ __forceinline__ float4 lrg_ld_coh4(const float *, unsigned) { return make_float4(0.f, 0.f, 0.f, 0.f); }
#endif
__device__ __forceinline__ void lrg_st_coh2(float *p, float a, float b) {      // 8-byte aligned
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Weight operands are software-pipelined ACROSS passes (a pass = one layer x one 128-column block): a ring of FD
// k-groups of B registers is always FD groups ahead of the MFMAs, and during the last FD groups of a pass it is
// refilled with the first FD groups of the NEXT pass (next column block or next layer), so the L2 latency of a pass's
// first weights hides behind the previous pass's MFMAs, epilogue and barrier instead of stalling every pass.
// The weights come pre-arranged in operand order (lrg_pack_weights): ONE global_load_dwordx4 per lane per k-group --
// measured (tools/mfma_peak.hip) 92 % of the fp32 MFMA peak against 78-86 % with four strided dword loads per group,
// whose address arithmetic and issue slots compete with the MFMAs.
template <int FD>
__device__ __forceinline__ void prefetch_b(float4 (&bq)[FD], const float4 *wp) {
#pragma unroll
    for (int g = 0; g < FD; ++g) bq[g] = wp[g * 64];
}

// A (32*RTT)x32 output strip (RTT 32x32 tiles stacked in rows) over NG k-groups of 8.  Lane half h feeds logical
// k = 8g + 4h + s of both operands:
//   A (activations) from LDS, one ds_read_b128 per tile per group, two groups ahead;
//   B (weights) from L2 into the register ring, shared by the RTT tiles.  No wave shares its B columns with another
//   wave, so an LDS round trip would buy nothing and its barriers would serialise the waves.
//   wp / wpn = this lane's float4 of group 0 of this / the next pass.
template <int NG, int RT, int RTT, int FD>
__device__ __forceinline__ void tile_mfma(f32x16 (&acc)[RT], const float *ap, int ld_in, const float4 *wp,
                                          const float4 *wpn, float4 (&bq)[FD]) {
    static_assert(NG >= FD && NG >= 2 && NG % FD == 0, "the ring must not be deeper than a pass, and a pass must leave it where the next one expects its first groups");
    float4 ar[3][RTT];                  // A operands of groups g, g+1, g+2 (explicit rotation: program order = issue order)
#pragma unroll
    for (int t = 0; t < RTT; ++t) {
        ar[0][t] = *reinterpret_cast<const float4 *>(ap + t * 32 * ld_in);
        ar[1][t] = *reinterpret_cast<const float4 *>(ap + t * 32 * ld_in + 8);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 2 < NG)
#pragma unroll
            for (int t = 0; t < RTT; ++t) ar[(g + 2) % 3][t] = *reinterpret_cast<const float4 *>(ap + t * 32 * ld_in + 8 * (g + 2));
        const float4 b = bq[g % FD];
#pragma unroll
        for (int t = 0; t < RTT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[g % 3][t].x, b.x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RTT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[g % 3][t].y, b.y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RTT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[g % 3][t].z, b.z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RTT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[g % 3][t].w, b.w, acc[t], 0, 0, 0);
#ifndef LRG_EXP_NO_WEIGHT_STREAM      // (experiment switch, --policy gt only: the ring is never refilled -- what streaming every tile's weights from L2 costs at saturation)
        bq[g % FD] = (g + FD < NG) ? wp[(g + FD) * 64] : wpn[(g + FD - NG) * 64];
#endif
    }
    // pin that order: two groups of LDS reads up front, then per k-group [LDS reads of g+2][4*RTT MFMAs][ring refill]
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * RTT, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 2 < NG) __builtin_amdgcn_sched_group_barrier(0x100, RTT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * RTT, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
}

// A narrow first layer (K = 13 -> two k-groups; the packed image and the staged rows are both zero-padded): its
// weights do not go through the ring; bf holds the first two groups, fetched before the input rows were staged.
template <int RT, int RTT>
__device__ __forceinline__ void tile_mfma_first(f32x16 (&acc)[RT], const float *ap, int ld_in, const float4 *wp, int ng,
                                                const float4 (&bf)[2], bool pre) {
    for (int g = 0; g < ng; ++g) {
        float4 a[RTT];
#pragma unroll
        for (int t = 0; t < RTT; ++t) a[t] = *reinterpret_cast<const float4 *>(ap + t * 32 * ld_in + 8 * g);
        const float4 b = (pre && g == 0) ? bf[0] : (pre && g == 1) ? bf[1] : wp[g * 64];
#pragma unroll
        for (int t = 0; t < RTT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b.w, acc[t], 0, 0, 0);
        }
    }
}

// A 64-wide layer of a single 32-row tile on ALL four waves: wave w computes columns 16 w .. 16 w + 15 as two 16 x 16 blocks (rows 0-15, 16-31) with
// v_mfma_f32_16x16x4_f32.  As 32 x 32 strips only two of the four waves have columns in such a layer (2.7 k of MFMAs + epilogue + barrier = ~5 k cycles per
// layer, and a branch tile without its two 64 -> 64 layers makes a step 7 us shorter: profiles/r04_knockout_gt.txt).
// Same bits as tile_mfma: that instruction adds its four products as an FMA chain over its lane groups q = 0 .. 3 (tools/mfma16_probe.hip: 100 % of 131 072 outputs),
// so with lane group q of instruction j fed k = 8 g + {0, 4, 1, 5 | 2, 6, 3, 7}[4 j + q] an output sees the chain k = 8 g + 0, 4, 1, 5, 2, 6, 3, 7 of the 32 x 32 x 2
// formulation (lane half h feeds k = 8 g + 4 h + s in instruction s).
//   A: lane (i = lane % 16, q) reads k half q & 1 of rows i and 16 + i (two ds_read_b128 per k-group) and uses x, z (q < 2) or y, w (q >= 2) of each;
//   B: the packed image as it is: the lane's float4 of (k half q & 1, column 16 w + i) holds k = 8 g + 4 (q & 1) + 0 .. 3 -- x, z for q < 2, y, w for q >= 2.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int NG, int FD>
__device__ __forceinline__ void tile_mfma16_n64(f32x4v (&acc)[2], const float *ap, int ld16, const float4 *wp, const float4 *wpn, float4 (&bq)[FD], bool lower) {
    static_assert(NG >= FD && NG >= 2 && NG % FD == 0, "the ring must not be deeper than a pass, and a pass must leave it where the next one expects its first groups");
    // ap: row i = lane % 16, k half q & 1; ap + ld16: the same of row 16 + i.  Two ds_read_b128 per k-group (lane groups q and q ^ 2 read the same words:
    // broadcast) -- v_permlane32_swap exchanges of one read's halves were measured first and cost ~500 cycles per k-group.
    // A k-group is 4 x 32 cycles of MFMAs here, not 4 x 64: the ring's FD groups of lead are ~500 cycles, less than a trip to L2 -- the weights of the groups
    // behind the ring's (FD .. NG - 1) are requested at once, up front, and the ring is refilled for the NEXT pass only.
    float4 bx[NG - FD];
#pragma unroll
    for (int g = FD; g < NG; ++g) bx[g - FD] = wp[g * 64];
    float4 ar[3][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        ar[0][r] = *reinterpret_cast<const float4 *>(ap + r * ld16);
        ar[1][r] = *reinterpret_cast<const float4 *>(ap + r * ld16 + 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 2 < NG)
#pragma unroll
            for (int r = 0; r < 2; ++r) ar[(g + 2) % 3][r] = *reinterpret_cast<const float4 *>(ap + r * ld16 + 8 * (g + 2));
        const float4 a0 = ar[g % 3][0], a1 = ar[g % 3][1];
        const float4 b = g < FD ? bq[g] : bx[g - FD];
        const float b0 = lower ? b.x : b.y, b1 = lower ? b.z : b.w;
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(lower ? a0.x : a0.y, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(lower ? a1.x : a1.y, b0, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(lower ? a0.z : a0.w, b1, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(lower ? a1.z : a1.w, b1, acc[1], 0, 0, 0);
        if (g < FD) bq[g] = wpn[g * 64];          // (the next pass's first FD groups, as every pass leaves them)
        // (nothing moves across a k-group: left to itself the compiler sinks the loads to their uses)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// LDS floats of one tile
#define LRG_TILE_LDS_FLOATS(CAP0, CAP1, RT, PACKED) ((CAP0) + (CAP1) + 512 + ((PACKED) ? 3 * 32 * (RT) + 8 : 0))

#ifndef LRG_TRACE
#define LRG_TRACE 0     // = CAP0 of the instantiation to trace (4352 / 2176 branch, 8320 head): thread 0 of each workgroup stamps the cycle counter at phase boundaries
#endif
#if LRG_TRACE
// stamps are parked in LDS and written out once at the end: a global store per stamp would sit in the same in-order memory
// counter as the weight loads and stretch the very phases it measures (~1.6 k cycles per store, seen)
#define TRACE(i) do { if (CAP0 == LRG_TRACE && tid == 0 && trace_sh) trace_sh[(i)] = (long long)__builtin_readcyclecounter(); } while (0)
#ifndef LRG_TRACE_LAYER
#define LRG_TRACE_LAYER 4   // the layer whose passes are stamped one by one (slots 12 + 2 * pass: after the MFMAs, 13 + 2 * pass: after the epilogue)
#endif
#define TRACE_PASS(l, cb, k) do { if ((l) == LRG_TRACE_LAYER && (cb) < 4) TRACE(12 + 2 * (cb) + (k)); } while (0)
#else
#define TRACE(i)
#define TRACE_PASS(l, cb, k)
#endif

// DIRECT: also compile the register-to-HBM copy of layers that do not stay in LDS (LRG_FWD_KEEP_ACTS on the pooled layer
// and on an in-place head layer; parity tests only) -- it costs ~25 VGPRs, which is the third wave per SIMD.
// PACKED: the rows of all instances are stored back to back (only the distinct ones, lrg_front_kernel); a tile is 32
// consecutive packed rows and may hold rows of several instances -- the runs of equal row_inst inside it.
//   r0 = first row of the tile in P.x; inst = the tile's instance (not PACKED); nvalid = rows of the instance that are not copies
//   (not PACKED; INT_MAX: all); nrows_packed = packed rows in all (PACKED); trace_sh: LDS stamps of an LRG_TRACE build (or null).
// Returns the number of runs (1 unless PACKED), 0 for a tile that was skipped.
// ONE (with PACKED): all rows of the tile belong to ONE instance, `inst`, and all of them are live (the free-running kernel: a slot's
// rows have a place of their own, padded to whole tiles with copies of its last row) -- no run detection, the centre and the
// per-instance bias addressed by `inst` directly (no trip through row_inst first).
#ifdef LRG_EXP_NO_TILE_SYNC      // (experiment switch, --policy gt only: what the LDS barriers INSIDE a tile cost -- a tile's results are garbage without them)
#define LRG_TILE_SYNC(t) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define LRG_TILE_SYNC(t) (t).sync()
#endif
struct LrgNoWait { static constexpr bool late = false; __device__ __forceinline__ void operator()() const {} };
// before_inst_bias: called by every thread right before the first per-instance bias (the hoisted pooled product of a head) is read --
// the free-running kernel's head tiles wait there for the pooled product of their slot.  WAIT::late (ONE only): the first pass of that
// layer runs its MFMAs first and waits (and fetches its bias values) in front of its epilogue -- a tile that was started before the
// pooled product is complete has the staging and a pass of MFMAs to do meanwhile.
// nrows_out (COH heads): only the logits of the tile's first nrows_out rows are stored -- a slot's TAIL rows (fewer than 32) sit in rows that slots share
// (lrg_async.inl, "shared tail tiles"); the rows behind them are other slots', whose logits this tile must not touch.
// part / nparts (ONE only, nparts 1, 2 or 4): the column blocks of the POOLED layer -- 128 -> 512: four passes, more than half of a branch
// tile's time -- are shared among nparts tasks that each run the layers before it again (a quarter of the FLOPs); the column maxima
// are independent of each other, part 0 alone stores conv[1].  A tile's latency for its work, where teams are idle anyway.
template <int CAP0, int CAP1, int RT, int FD, bool DIRECT, bool PACKED, bool COH, class TEAM, bool ONE = false, class WAIT = LrgNoWait>
__device__ __forceinline__ int lrg_fused_tile(const LrgFusedProb &P, long r0, int inst, int tile, int nvalid, int nrows_packed,
                                              float *smem, const TEAM &team, long long *trace_sh, const WAIT &before_inst_bias = WAIT(),
                                              int part = 0, int nparts = 1, int nrows_out = 32 * RT) {
    constexpr int FM = 32 * RT;      // rows (points) per tile
    static_assert(!PACKED || RT == 1, "packed rows use 32-row tiles");
    float *buf0 = smem;                       // outputs of even layers
    float *buf1 = smem + CAP0;                // the staged input and outputs of odd layers
    float *poolbuf = smem + CAP0 + CAP1;      // [512] running column maxima of the pooled layer / final-layer weights
    int *run_start = reinterpret_cast<int *>(poolbuf + 512);   // PACKED: [FM + 1] first row of each run (and the end)
    int *run_inst = run_start + FM + 1;                        //         [FM] instance of each run, -1 = dead rows past *nrows
    int *run_count = run_inst + FM;                            //         [1]

    const int tid = team.tid();
    const int lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
#if LRG_TRACE
    if (CAP0 == LRG_TRACE && tid < 32 && trace_sh) trace_sh[tid] = 0;
#endif
    TRACE(0);

    // A 64-wide layer of a 64-row tile is laid out 2x2 (each wave one 32x32 tile) instead of 1x4 strips of which only
    // two would have columns: all four SIMDs stay busy through the narrow layers.
    auto is22 = [&](const LrgFusedLayer &L, int l) { return RT == 2 && L.N == 64 && (L.K == 64 || l == 0); };
    // ... and a 64 -> 64 layer of a single-instance 32-row tile runs on 16 x 16 x 4 MFMAs, every wave 16 columns (tile_mfma16_n64)
    auto isn16 = [&](const LrgFusedLayer &L) {
        return LRG_TILE_N16 && PACKED && ONE && RT == 1 && L.N == 64 && L.K == 64 && (L.flags & (LRG_FL_RELU | LRG_FL_KEEP)) == (LRG_FL_RELU | LRG_FL_KEEP) &&
               !(L.flags & (LRG_FL_POOL | LRG_FL_INST_BIAS | LRG_FL_INPLACE));
    };
    auto col_of = [&](const LrgFusedLayer &L, int l, int cb) { return is22(L, l) ? (wn & 1) * 32 : isn16(L) ? wn * 16 : cb * FBN + wn * 32; };

    // this lane's float4 of k-group 0 of the 32-column block starting at column c of layer L (packed image)
    auto wptr = [&](const LrgFusedLayer &L, int c) {
        if (isn16(L))       // (lane group q reads k half q & 1 of its column 16 wn + lane % 16)
            return reinterpret_cast<const float4 *>(L.w) + (long)(c >> 5) * L.ng * 64 + ((lane >> 4) & 1) * 32 + ((c + (lane & 15)) & 31);
        return reinterpret_cast<const float4 *>(L.w) + (long)(c >> 5) * L.ng * 64 + lane;
    };
    // bias of column c of layer L for this lane (a per-instance row when the layer carries the hoisted pooled product)
    auto bias_of = [&](const LrgFusedLayer &L, int c) -> float {
        if (!L.bias) return 0.f;
        if (PACKED && (L.flags & LRG_FL_INST_BIAS)) return 0.f;       // added per run of rows in the epilogue
        if (isn16(L)) return L.bias[c + (lane & 15)];
        return (L.flags & LRG_FL_INST_BIAS) ? L.bias[(long)inst * L.N + c + li] : L.bias[c + li];
    };
    float4 bq[FD], bf[2];
    float bvn;                                   // bias of the NEXT pass, fetched one pass ahead like the weights
    LrgFusedLayer Lnext = P.L[0];
    {   // the first pass's weights start their trip before the input rows are staged
        int c = col_of(Lnext, 0, 0);
        if (c >= Lnext.N) c = 0;
        bvn = bias_of(Lnext, c);
        const float4 *wp0 = wptr(Lnext, c);
        const int K0 = Lnext.K;
        if (K0 == 64 || K0 == 128 || K0 == 256) prefetch_b<FD>(bq, wp0);
        else { bf[0] = wp0[0]; bf[1] = wp0[Lnext.ng > 1 ? 64 : 0]; }
    }

    // ---- stage the input rows into buf1, zero-padded to a multiple of 8 columns ----
    const int Kin = P.Kin;
    const int Kp = (Kin + 7) & ~7;
    const int ld_x = Kp + 4;
    if ((P.ldx & 3) == 0 && (Kin & 3) == 0 && (((uintptr_t)P.x) & 15) == 0 && !(PACKED && P.center)) {
        const int q = Kp >> 2;
        for (int idx = tid; idx < FM * q; idx += FTHREADS) {
            int row = idx / q, c4 = idx - row * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * c4 < Kin) {
                if constexpr (COH) v = lrg_ld_coh4(P.x + r0 * P.ldx, (unsigned)(row * P.ldx + 4 * c4) * 4u);
                else v = *reinterpret_cast<const float4 *>(P.x + (r0 + row) * P.ldx + 4 * c4);
            }
            *reinterpret_cast<float4 *>(&buf1[row * ld_x + 4 * c4]) = v;
        }
    } else if (PACKED && ONE && COH && P.center && P.ldx == 16 && Kp == 16) {
        // rows at a 64-byte stride (the free-running kernel's gather, LrgFrontArgs.rows16): four 16-byte loads per row, the centre's sixteen
        // floats likewise; columns past Kin are zeros on both sides
        for (int idx = tid; idx < FM * 4; idx += FTHREADS) {
            const int row = idx >> 2, c4 = idx & 3;
            const float4 xv = lrg_ld_coh4(P.x + r0 * 16, (unsigned)(row * 16 + 4 * c4) * 4u);
            const float4 cv = lrg_ld_coh4(P.center + (long)inst * 16, (unsigned)(4 * c4) * 4u);
            *reinterpret_cast<float4 *>(&buf1[row * ld_x + 4 * c4]) = make_float4(__fsub_rn(xv.x, cv.x), __fsub_rn(xv.y, cv.y), __fsub_rn(xv.z, cv.z), __fsub_rn(xv.w, cv.w));
        }
    } else if (PACKED && !ONE && COH && P.center && P.ldx == 16 && Kp == 16) {
        // a shared tile of the free-running kernel (rows at a 64-byte stride, of several slots): the row's tag first, then its centre -- two round trips for
        // the tile instead of three per element of the general form below
        for (int idx = tid; idx < FM * 4; idx += FTHREADS) {
            const int row = idx >> 2, c4 = idx & 3;
            const float4 xv = lrg_ld_coh4(P.x + r0 * 16, (unsigned)(row * 16 + 4 * c4) * 4u);
            const int ins = (r0 + row < nrows_packed) ? lrg_ld_coh(P.row_inst + r0 + row) : -1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ins >= 0) {
                const float4 cv = lrg_ld_coh4(P.center + (long)ins * 16, (unsigned)(4 * c4) * 4u);
                v = make_float4(__fsub_rn(xv.x, cv.x), __fsub_rn(xv.y, cv.y), __fsub_rn(xv.z, cv.z), __fsub_rn(xv.w, cv.w));
            }
            *reinterpret_cast<float4 *>(&buf1[row * ld_x + 4 * c4]) = v;
        }
    } else if (PACKED && P.center) {
        // uncentred rows: subtract the owning instance's centre while staging (same float32 subtraction the gather would do)
        for (int idx = tid; idx < FM * Kp; idx += FTHREADS) {
            int row = idx / Kp, c = idx - row * Kp;
            float v = 0.f;
            if (c < Kin) {
                const int ins = ONE ? inst : (r0 + row < nrows_packed) ? (COH ? lrg_ld_coh(P.row_inst + r0 + row) : P.row_inst[r0 + row]) : -1;
                if constexpr (COH) {
                    const float xv = lrg_ld_coh(P.x + (r0 + row) * P.ldx + c);
                    v = ins >= 0 ? __fsub_rn(xv, lrg_ld_coh(P.center + ins * 16 + c)) : 0.f;
                } else {
                    const float xv = P.x[(r0 + row) * P.ldx + c];
                    v = ins >= 0 ? __fsub_rn(xv, P.center[ins * 16 + c]) : 0.f;
                }
            }
            buf1[row * ld_x + c] = v;
        }
    } else {
        for (int idx = tid; idx < FM * Kp; idx += FTHREADS) {
            int row = idx / Kp, c = idx - row * Kp;
            buf1[row * ld_x + c] = c < Kin ? (COH ? lrg_ld_coh(P.x + (r0 + row) * P.ldx + c) : P.x[(r0 + row) * P.ldx + c]) : 0.f;
        }
    }
    // poolbuf: running column maxima of a pooled stack, or the final [C,2] layer of a head (C <= 256)
    if (P.fw) { for (int i = tid; i < 2 * P.L[P.nlayers - 1].N; i += FTHREADS) poolbuf[i] = P.fw[i]; }
    else { for (int i = tid; i < 512; i += FTHREADS) poolbuf[i] = 0.f; }      // (PACKED keeps its maxima elsewhere)
    if (tile * FM >= nvalid) return 0;             // uniform over the tile's threads
    if (PACKED && ONE) {
        if (tid == 0) { run_start[0] = 0; run_start[1] = FM; run_inst[0] = inst; *run_count = 1; }
    } else if (PACKED && tid < 64) {
        // runs of equal instance among the tile's rows (rows past *nrows: instance -1), found by wave 0 with one ballot
        const int row = tid & 31;
        const int mine = (r0 + row < nrows_packed) ? (COH ? lrg_ld_coh(P.row_inst + r0 + row) : P.row_inst[r0 + row]) : -1;
        const int prev = __shfl_up(mine, 1);
        const bool start = tid < 32 && (row == 0 || mine != prev);
        const unsigned long long m = __ballot(start);
        if (start) {
            const int k = __popcll(m & ((1ull << row) - 1ull));
            run_start[k] = row;
            run_inst[k] = mine;
        }
        if (tid == 0) { const int n = __popcll(m); run_start[n] = FM; *run_count = n; }
    }
    LRG_TILE_SYNC(team);
    TRACE(1);
    const int nruns = (PACKED && !ONE) ? *run_count : 1;

    const int nlayers = P.nlayers;
    int prevN = Kp;
    int lastN = 0, lastflags = 0;
    for (int l = 0; l < nlayers; ++l) {
        const LrgFusedLayer L = Lnext;             // descriptors are fetched one layer ahead (scalar loads off the critical path)
        if (l + 1 < nlayers) Lnext = P.L[l + 1];   // (after the last layer Lnext == L: the ring refill stays in bounds)
#ifdef LRG_EXP_SKIP_NARROW      // (experiment switch, --policy gt only: a branch tile without its two 64 -> 64 layers)
        if (ONE && nlayers == 5 && (l == 1 || l == 2)) { prevN = L.N; continue; }
#endif
        const bool inplace = (L.flags & LRG_FL_INPLACE) != 0;
        const float *act_in = (l & 1) ? buf0 : buf1;
        float *act_out = ((l & 1) != 0) == !inplace ? buf1 : buf0;
        const int ld_in = prevN + 4, ld_out = L.N + 4;
        const bool m22 = is22(L, l);
        const int rbase = m22 ? (wn >> 1) * 32 : 0;          // first row of this wave's strip within the tile
        const int ntile = m22 ? 1 : RT;
        const float *ap = act_in + (rbase + li) * ld_in + 4 * lh;
        const bool n16 = isn16(L);
        const int ncb = (m22 || n16) ? 1 : (L.N + FBN - 1) / FBN;
        // (this task's share of the pooled layer's column blocks; every other layer whole)
        const bool shared = ONE && nparts > 1 && (L.flags & LRG_FL_POOL) && !(L.flags & LRG_FL_KEEP);
#ifdef LRG_EXP_HALF_POOL      // (experiment switch, --policy gt only: a branch tile without the second half of its pooled layer's column blocks)
        const int cb_lo = shared ? part * ncb / nparts : 0,
                  cb_hi = (ONE && nlayers == 5 && (L.flags & LRG_FL_POOL) && !(L.flags & LRG_FL_KEEP) && !shared) ? ncb / 2 : shared ? (part + 1) * ncb / nparts : ncb;
#else
        const int cb_lo = shared ? part * ncb / nparts : 0, cb_hi = shared ? (part + 1) * ncb / nparts : ncb;
#endif
        TRACE(2 + 2 * l);
        for (int cb = cb_lo; cb < cb_hi; ++cb) {
            const int col0 = col_of(L, l, cb);
            const bool wave_on = col0 < L.N;         // a 64-wide layer outside the 2x2 layout keeps two of the four waves busy
            // the pass after this one: next column block, else the next layer's first
            const bool same = cb + 1 < cb_hi;
            const LrgFusedLayer &Lx = same ? L : Lnext;
            // (the first pass of the next layer: this task's first column block of it, if that is the shared pooled layer)
            const bool next_shared = !same && ONE && nparts > 1 && (Lx.flags & LRG_FL_POOL) && !(Lx.flags & LRG_FL_KEEP) && l + 1 < nlayers;
            const int cb_next = same ? cb + 1 : next_shared ? part * ((Lx.N + FBN - 1) / FBN) / nparts : 0;
            int coln = col_of(Lx, same ? l : l + 1, cb_next);
            if (coln >= Lx.N) coln = 0;
            const float4 *wpn = wptr(Lx, coln);

            if constexpr (PACKED && ONE && RT == 1) {
                if (n16) {
                    const float bv = bvn;
                    bvn = bias_of(Lx, coln);
                    f32x4v c2[2];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { c2[0][t] = 0.f; c2[1][t] = 0.f; }
                    const int i16 = lane & 15, q4 = lane >> 4;
                    tile_mfma16_n64<8, FD>(c2, act_in + i16 * ld_in + 4 * (q4 & 1), 16 * ld_in, wptr(L, col0), wpn, bq, lane < 32);
                    // the block of rows 16 r .. 16 r + 15: lane (i16, q4) holds rows 16 r + 4 q4 + t of column col0 + i16
                    float *o = act_out + (4 * q4) * ld_out + col0 + i16;
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int t = 0; t < 4; ++t) o[(16 * r + t) * ld_out] = fmaxf(c2[r][t] + bv, 0.f);
                    continue;
                }
            }

            f32x16 acc[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
            const float bv = bvn;
            bvn = bias_of(Lx, coln);
            // PACKED: the per-instance bias values of the first RB runs start their trip before the MFMAs (a load that the
            // epilogue issues and waits for costs a full memory round trip per run and pass: the rows were written by the
            // GEMM kernel on another XCD a moment ago)
            constexpr int RB = 4;
            float bk[RB] = {0.f, 0.f, 0.f, 0.f};
            bool late_bias = false;
            if constexpr (PACKED) {
                if ((L.flags & LRG_FL_INST_BIAS) && L.bias && cb == cb_lo) {
                    if constexpr (WAIT::late && ONE) late_bias = true;
                    else before_inst_bias();
                }
                if (wave_on && (L.flags & LRG_FL_INST_BIAS) && L.bias && !late_bias) {
#pragma unroll
                    for (int k = 0; k < (ONE ? 1 : RB); ++k) {
                        const int ins = ONE ? inst : k < nruns ? run_inst[k] : -1;
                        if (ins >= 0) bk[k] = COH ? lrg_ld_coh(L.bias + (long)ins * L.N + col0 + li) : L.bias[(long)ins * L.N + col0 + li];
                    }
                }
            }
            if (wave_on) {
                const float4 *wp = wptr(L, col0);
                if constexpr (RT == 2) {
                    if (m22) {
                        if (L.K == 64) tile_mfma<8, RT, 1, FD>(acc, ap, ld_in, wp, wpn, bq);
                        else {
                            prefetch_b<FD>(bq, wpn);
                            tile_mfma_first<RT, 1>(acc, ap, ld_in, wp, L.ng, bf, cb == cb_lo);
                        }
                    } else if (L.K == 128) tile_mfma<16, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else if (L.K == 64) tile_mfma<8, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else if (L.K == 256) tile_mfma<32, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else {
                        prefetch_b<FD>(bq, wpn);
                        tile_mfma_first<RT, RT>(acc, ap, ld_in, wp, L.ng, bf, cb == cb_lo);
                    }
                } else {
                    if (L.K == 128) tile_mfma<16, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else if (L.K == 64) tile_mfma<8, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else if (L.K == 256) tile_mfma<32, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else {
                        prefetch_b<FD>(bq, wpn);
                        tile_mfma_first<RT, RT>(acc, ap, ld_in, wp, L.ng, bf, cb == cb_lo);
                    }
                }
            } else {
                prefetch_b<FD>(bq, wpn);      // an idle wave still owes the next pass its first weights
            }
            TRACE_PASS(l, cb, 0);
            if (inplace) LRG_TILE_SYNC(team);                // the output overlays this layer's input: everyone must be done reading
            if constexpr (PACKED && ONE && WAIT::late) {
                if (late_bias) {
                    before_inst_bias();
                    if (wave_on) bk[0] = COH ? lrg_ld_coh(L.bias + (long)inst * L.N + col0 + li) : L.bias[(long)inst * L.N + col0 + li];
                }
            }
            if (wave_on) {
                // ---- epilogue: bias, ReLU, keep in LDS / copy to HBM / column max ----
                const int col = col0 + li;
                float cmax = 0.f;
                // a layer that stays in LDS is copied to HBM from there after the barrier (coalesced); only the
                // parity-test copy of a layer that does not (KEEP_ACTS on the pooled layer) is stored from registers
                float *gdirect = (DIRECT && L.gout && (!(L.flags & LRG_FL_KEEP) || inplace)) ? L.gout + r0 * L.N : nullptr;
                if constexpr (PACKED) {
                    // One wave per SIMD: every instruction of the epilogue is issue time the matrix pipe idles through
                    // (~600 instructions = 2.6 k cycles per pass against 4.1 k of MFMAs, profiles/r02_branch_pass_stamps.txt),
                    // so this path is written for instruction count: flag tests outside the 16-value loops, one unsigned
                    // compare per value for "row in run", the single-run tile (4 of 5) without any row test.
                    f32x16 &a = acc[0];
                    const bool relu = (L.flags & LRG_FL_RELU) != 0, keep = (L.flags & LRG_FL_KEEP) != 0;
                    const int row4 = 4 * lh;
                    // per-instance bias (the hoisted pooled product of a head, :128-141): one value per run of rows; the
                    // first RB runs' values were requested before the MFMAs of this pass
                    if ((L.flags & LRG_FL_INST_BIAS) && L.bias) {
                        if (nruns == 1) {
#pragma unroll
                            for (int rr = 0; rr < 16; ++rr) a[rr] += bk[0];
                        } else {
                            auto add_run = [&](int k, float b) {
                                const int lo = run_start[k], len = run_start[k + 1] - lo, d = row4 - lo;
#pragma unroll
                                for (int rr = 0; rr < 16; ++rr)
                                    if ((unsigned)(d + (rr & 3) + 8 * (rr >> 2)) < (unsigned)len) a[rr] += b;
                            };
#pragma unroll
                            for (int k = 0; k < RB; ++k)
                                if (k < nruns) add_run(k, bk[k]);
                            for (int k = RB; k < nruns; ++k) {
                                const int ins = run_inst[k];
                                add_run(k, ins >= 0 ? (COH ? lrg_ld_coh(L.bias + (long)ins * L.N + col) : L.bias[(long)ins * L.N + col]) : 0.f);
                            }
                        }
                    }
                    if (relu) {
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) a[rr] = fmaxf(a[rr] + bv, 0.f);
                    } else {
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) a[rr] += bv;
                    }
                    if (keep) {
                        float *o = act_out + row4 * ld_out + col;
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) o[((rr & 3) + 8 * (rr >> 2)) * ld_out] = a[rr];
                    }
                    if (DIRECT && gdirect) {
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) gdirect[(unsigned)((row4 + (rr & 3) + 8 * (rr >> 2)) * L.N + col)] = a[rr];
                    }
                    if (L.flags & LRG_FL_POOL) {
                        // Column maxima per run of rows.  The values are >= 0, so the maximum is taken on their bit patterns
                        // as integers (v_max3_i32, no NaN canonicalisation; the same order the atomicMax below relies on).
                        // They are parked in the layer's own output buffer (free: the pooled layer does not stay in LDS) and
                        // go to the instances' pooled features after the layer, so that no pass carries an atomic's
                        // memory-side round trip in the in-order counter its weight loads use.
                        const int runcap = keep ? 0 : (act_out == buf1 ? CAP1 : CAP0) / L.N;
                        int *parked = reinterpret_cast<int *>(act_out);
                        auto put = [&](int k, int m) {
                            const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)m, (unsigned)m, false, false);
                            m = max((int)sw[0], (int)sw[1]);                 // both halves of the wave hold rows of the column
                            if (lh == 0) {
                                if (k < runcap) parked[k * L.N + col] = m;
                                else {
                                    const int ins = ONE ? inst : run_inst[k];
                                    if (ins >= 0 && m > 0) atomicMax(reinterpret_cast<int *>(P.pool + (long)ins * P.pool_stride + col), m);
                                }
                            }
                        };
                        if (nruns == 1) {
                            int m = 0;
#pragma unroll
                            for (int rr = 0; rr < 16; ++rr) m = max(m, __float_as_int(a[rr]));
                            put(0, m);
                        } else {
                            for (int k = 0; k < nruns; ++k) {
                                const int lo = run_start[k], len = run_start[k + 1] - lo, d = row4 - lo;
                                int m = 0;
#pragma unroll
                                for (int rr = 0; rr < 16; ++rr)
                                    m = max(m, (unsigned)(d + (rr & 3) + 8 * (rr >> 2)) < (unsigned)len ? __float_as_int(a[rr]) : 0);
                                put(k, m);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        if (t < ntile) {
#pragma unroll
                            for (int rr = 0; rr < 16; ++rr) {
                                const int rl = rbase + t * 32 + 4 * lh + (rr & 3) + 8 * (rr >> 2);
                                float v = acc[t][rr] + bv;
                                if (L.flags & LRG_FL_RELU) v = fmaxf(v, 0.f);
                                if (L.flags & LRG_FL_KEEP) act_out[rl * ld_out + col] = v;
                                if (DIRECT && gdirect) gdirect[(unsigned)(rl * L.N + col)] = v;
                                cmax = fmaxf(cmax, v);
                            }
                        }
                    }
                    if (L.flags & LRG_FL_POOL) {
                        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
                        if (lh == 0) {
                            if (m22) atomicMax(reinterpret_cast<int *>(&poolbuf[col]), __float_as_int(cmax));   // two waves share the column (values >= 0)
                            else poolbuf[col] = fmaxf(poolbuf[col], cmax);                                      // this wave owns the column
                        }
                    }
                }
            }
            TRACE_PASS(l, cb, 1);
        }
        LRG_TILE_SYNC(team);                                 // layer boundary: outputs visible, inputs dead
        if constexpr (PACKED) {
            if (ONE && COH && P.pool_rows && (L.flags & LRG_FL_POOL) && !(L.flags & LRG_FL_KEEP)) {
                // the tile's maxima as one row for whoever takes the maximum over the slot's tiles (the pooled-product units): 16 bytes
                // per lane, write-through -- N atomics per tile were N write transactions and N acknowledgements to drain before the arrival
                float *dst = P.pool_rows + (long)inst * P.pool_rows_stride + (long)tile * L.N;
                for (int c4 = (cb_lo * FBN >> 2) + tid; c4 < (min(L.N, cb_hi * FBN) >> 2); c4 += FTHREADS)
                    lrg_st_coh4(dst, (unsigned)c4 * 16u, *reinterpret_cast<const float4 *>(act_out + 4 * c4));
            } else if ((L.flags & LRG_FL_POOL) && !(L.flags & LRG_FL_KEEP)) {
                // the parked per-run maxima -> the pooled features (:122-125), coalesced, nothing waits for them
                const int runcap = (act_out == buf1 ? CAP1 : CAP0) / L.N;
                const int nk = nruns < runcap ? nruns : runcap;
                for (int k = 0; k < nk; ++k) {
                    const int ins = ONE ? inst : run_inst[k];
                    if (ins < 0) continue;
                    int *dst = reinterpret_cast<int *>(P.pool + (long)ins * P.pool_stride);
#ifndef LRG_EXP_NO_POOL_ATOMICS      // (experiment switch: what the atomics cost, measured under --policy gt where the logits do not steer the growth)
                    for (int c = cb_lo * FBN + tid; c < min(L.N, cb_hi * FBN); c += FTHREADS) {      // (the columns this task ran)
                        const int m = reinterpret_cast<const int *>(act_out)[k * L.N + c];
                        if (m > 0) atomicMax(dst + c, m);
                    }
#endif
                }
            }
        }
        if (L.gout && (L.flags & LRG_FL_KEEP) && !inplace && part == 0) {
            // HBM copy of a layer the next one reads from LDS (conv[1] for the heads, :130,:134): whole rows, float4
            const int q = L.N >> 2;
            float *gb = L.gout + r0 * L.N;
            // (a single-instance 32-row tile whose NEXT layer is 64 wide keeps two of its four waves busy there: the other two copy, the first two go on --
            //  the copy reads what the next layer reads; 640 of a branch tile's 52 k cycles, profiles/r04_delay_and_stamps.txt)
            const bool by_idle = PACKED && ONE && RT == 1 && l + 1 < nlayers && Lnext.N <= 64 && !(Lnext.flags & LRG_FL_INPLACE) && !isn16(Lnext);
            for (int idx = by_idle ? tid - FTHREADS / 2 : tid; idx < FM * q && idx >= 0; idx += by_idle ? FTHREADS / 2 : FTHREADS) {
                const int row = idx / q, c4 = idx - row * q;
                const float4 v = *reinterpret_cast<const float4 *>(act_out + row * ld_out + 4 * c4);
                if constexpr (COH) lrg_st_coh4(gb, (unsigned)(row * L.N + 4 * c4) * 4u, v);
                else *reinterpret_cast<float4 *>(gb + (unsigned)(row * L.N + 4 * c4)) = v;
            }
        }
        TRACE(2 + 2 * l + 1);
        prevN = L.N;
        lastN = L.N;
        lastflags = L.flags;
    }

    // ---- pooled maxima of this tile -> the instance's pooled feature (:122-125) ----
    if (!PACKED && (lastflags & LRG_FL_POOL) && P.pool) {
        float *dst = P.pool + (r0 / P.rows_per_inst) * P.pool_stride;
        for (int c = tid; c < lastN; c += FTHREADS) atomicMax(reinterpret_cast<int *>(&dst[c]), __float_as_int(poolbuf[c]));
    }
    // ---- final 2-wide layer of a head, no ReLU (:145-149, :158-162) ----
    if (P.fw) {
        const int C = lastN;
        const bool odd = ((nlayers - 1) & 1) != 0;
        const float *act = (odd == !(lastflags & LRG_FL_INPLACE)) ? buf1 : buf0;
        const int ld = C + 4;
        // FTHREADS / FM lanes per row, each taking every LPR-th float4 of the row; partial sums are combined by
        // xor-shuffles in a fixed order (deterministic).  The [C,2] weights were parked in LDS before the first barrier.
        constexpr int LPR = FTHREADS / FM;
        const int row = tid / LPR, q = tid % LPR;
        float s0 = 0.f, s1 = 0.f;
        for (int k = 4 * q; k < C; k += 4 * LPR) {
            const float4 a = *reinterpret_cast<const float4 *>(act + row * ld + k);
            const float4 w01 = *reinterpret_cast<const float4 *>(poolbuf + 2 * k);
            const float4 w23 = *reinterpret_cast<const float4 *>(poolbuf + 2 * k + 4);
            s0 = fmaf(a.x, w01.x, s0); s1 = fmaf(a.x, w01.y, s1);
            s0 = fmaf(a.y, w01.z, s0); s1 = fmaf(a.y, w01.w, s1);
            s0 = fmaf(a.z, w23.x, s0); s1 = fmaf(a.z, w23.y, s1);
            s0 = fmaf(a.w, w23.z, s0); s1 = fmaf(a.w, w23.w, s1);
        }
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) { s0 += __shfl_xor(s0, m); s1 += __shfl_xor(s1, m); }
        if constexpr (COH && LPR >= 2) {
            // two rows' logits per 16-byte write-through store: the lane of the even row takes the odd row's pair from its neighbour LPR lanes on
            const float t0 = s0 + P.fb[0], t1 = s1 + P.fb[1];
            const float u0 = __shfl_down(t0, LPR), u1 = __shfl_down(t1, LPR);
            if (q == 0 && !(row & 1)) {
                if (row + 1 < nrows_out) lrg_st_coh4(P.fout + r0 * 2, (unsigned)row * 8u, make_float4(t0, t1, u0, u1));
                else if (row < nrows_out) lrg_st_coh2(P.fout + (r0 + row) * 2, t0, t1);
            }
        } else if (q == 0) {
            if constexpr (COH) lrg_st_coh2(P.fout + (r0 + row) * 2, s0 + P.fb[0], s1 + P.fb[1]);
            else *reinterpret_cast<float2 *>(P.fout + (r0 + row) * 2) = make_float2(s0 + P.fb[0], s1 + P.fb[1]);
        }
    }
    // ---- leave the pooled feature of this instance zero for the next evaluation (it was consumed by the GEMV) ----
    if (!PACKED && P.zero_pool && tile == 0)
        for (int c = tid; c < P.zero_count; c += FTHREADS) P.zero_pool[(long)inst * P.zero_count + c] = 0.f;
    TRACE(20);
    return nruns;
}
