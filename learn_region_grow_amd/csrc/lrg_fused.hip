// Fused LrgNet stacks for gfx950: a whole branch (learn_region_grow_util.py:106-123) or a whole head (:138-162)
// per 64-row (branch) / 32-row (head) tile in ONE kernel.  Activations never leave the CU: they ping-pong between two
// LDS buffers; the layer weights (L2-resident, 3.2 MB in all, pre-arranged in operand order by lrg_pack_weights) go
// straight into the MFMA B operands through a register ring that runs a few k-groups ahead of the fp32 MFMAs
// (v_mfma_f32_32x32x2_f32, exact fp32) and never drains between layers.  Only conv[1] (needed by the heads), the
// pooled maxima and the logits are written to HBM; the 512-wide layer and the 1088-wide concat are never materialised.
//
// The layer-streamed formulation (one launch per layer, lrg_net.hip) stays available: it is what the layer-by-layer
// parity tests and the "HBM-streamed" roofline figure use.
#include "lrg_common.h"
#include "lrg_fused.h"
#include "lrg_median.h"

#ifndef LRG_TRACE
#define LRG_TRACE 0     // = CAP0 of the instantiation to trace (4352 / 2176 branch, 8320 head): thread 0 of each workgroup stamps the cycle counter at phase boundaries
#endif
#if LRG_TRACE
__device__ long long *g_lrg_trace = nullptr;
extern "C" void lrg_set_trace(long long *p) { hipMemcpyToSymbol(HIP_SYMBOL(g_lrg_trace), &p, sizeof(p)); }
// stamps are parked in LDS and written out once at the end: a global store per stamp would sit in the same in-order memory
// counter as the weight loads and stretch the very phases it measures (~1.6 k cycles per store, seen)
#define TRACE(i) do { if (CAP0 == LRG_TRACE && tid == 0) lrg_trace_sh[(i)] = (long long)__builtin_readcyclecounter(); } while (0)
#ifndef LRG_TRACE_LAYER
#define LRG_TRACE_LAYER 4   // the layer whose passes are stamped one by one (slots 12 + 2 * pass: after the MFMAs, 13 + 2 * pass: after the epilogue)
#endif
#define TRACE_PASS(l, cb, k) do { if ((l) == LRG_TRACE_LAYER && (cb) < 4) TRACE(12 + 2 * (cb) + (k)); } while (0)
#else
#define TRACE(i)
#define TRACE_PASS(l, cb, k)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));


#define FBN 128      // output columns per pass: 4 waves side by side, each a (32*RT)x32 strip (RT 32x32 MFMA tiles sharing B)
#define FTHREADS 256 // one wave per SIMD per workgroup; 2-3 workgroups per CU interleave without sharing barriers

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
    static_assert(NG >= FD && NG >= 2, "the ring must not be deeper than a pass");
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
        bq[g % FD] = (g + FD < NG) ? wp[(g + FD) * 64] : wpn[(g + FD - NG) * 64];
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

// One median workgroup of the packed branch launch (FTHREADS threads): LRG_MED_SPLIT workgroups per slot, workgroup g taking the
// centred channels g, g + SPLIT, g + 2 SPLIT, ...  Few workgroups: they sit in front of the tiles in the grid, and every CU slot
// they hold while the tiles are dealt out is a tile doubled up elsewhere (nine per slot: 56 us instead of 42).
// Up to 1024 points one wavefront per channel, keys in registers; above, the block bisection channel after channel.  A result
// goes out twice: plain (centre array, read by later launches) and as a tagged 64-bit word for the tile workgroups of THIS launch
// (one coherent load gives value and validity together).
#ifndef LRG_MED_SPLIT
#define LRG_MED_SPLIT 3
#endif
__device__ __forceinline__ void lrg_fused_median_wg(const LrgFusedMedians &M, int id, int *sh) {
    const int s = id / LRG_MED_SPLIT, g = id - s * LRG_MED_SPLIT, tid = threadIdx.x, wave = tid >> 6;
    if (M.big[2 * s] == 0) return;                                   // no rows this iteration (idle / finished slot)
    const long long t0 = M.phase_ticks ? wall_clock64() : 0;
    const unsigned tag = (unsigned)M.big[2 * s + 1];
    const LrgSlot *S = &M.slots[s];
    const LrgRoom *R = &M.rooms[S->room];
    const int F = M.F, nc = S->nc;
    const int32_t *idx = S->cur_idx;
    const float *points = R->points;
    auto publish = [&](int ch, float m) {
        M.center[s * 16 + ch] = m;
        __hip_atomic_store(&M.ctag[s * 16 + ch], ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(m),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (nc <= 1024) {
        const int y = g + LRG_MED_SPLIT * wave;
        const int ch = y < M.ncentred ? lrg_centred_channel(y, F) : -1;
        if (ch >= 0) {
            const float m = nc <= 256 ? lrg_median_wave_r<4>(points + ch, idx, F, nc) : lrg_median_wave_r<16>(points + ch, idx, F, nc);
            if ((tid & 63) == 0) publish(ch, m);
        }
    } else if (nc <= 16 * FTHREADS && LRG_MED_SPLIT == 3) {
        // the workgroup's three channels at once (shared index loads and barriers), radix select
        int chs[3];
        float m[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { const int y = g + 3 * c; chs[c] = y < M.ncentred ? lrg_centred_channel(y, F) : -1; }
        lrg_median_block_radix<16, FTHREADS, 3>(points, chs, idx, F, nc, sh, m);
        if (tid < 3 && chs[tid] >= 0) publish(chs[tid], tid == 0 ? m[0] : tid == 1 ? m[1] : m[2]);
    } else {
        for (int y = g; y < M.ncentred; y += LRG_MED_SPLIT) {
            const int ch = lrg_centred_channel(y, F);
            if (ch < 0) break;
            __syncthreads();
            float m;
            if (nc <= 48 * FTHREADS) {
                const int chs[1] = {ch};
                float mm[1];
                lrg_median_block_radix<48, FTHREADS, 1>(points, chs, idx, F, nc, sh, mm);
                m = mm[0];
            } else {
                const int k2 = nc >> 1, k1 = (nc & 1) ? k2 : k2 - 1;
                uint32_t ka, kb;
                lrg_select2(nullptr, false, points, idx, F, ch, nc, k1, k2, sh, &ka, &kb);
                const float lo = lrg_key2f(ka), hi = lrg_key2f(kb);
                m = (nc & 1) ? hi : __fmul_rn(__fadd_rn(lo, hi), 0.5f);
            }
            if (tid == 0) publish(ch, m);
        }
    }
    if (M.phase_ticks && tid == 0 && g == 0) M.phase_ticks[2 * s + 1] += wall_clock64() - t0;
}

// DIRECT: also compile the register-to-HBM copy of layers that do not stay in LDS (LRG_FWD_KEEP_ACTS on the pooled layer
// and on an in-place head layer; parity tests only) -- it costs ~25 VGPRs, which is the third wave per SIMD.
// PACKED: the rows of all instances are stored back to back (only the distinct ones, lrg_front_kernel); a tile is 32
// consecutive packed rows and may hold rows of several instances -- the runs of equal row_inst inside it.
template <int CAP0, int CAP1, int RT, int FD, int OCC, bool DIRECT, bool PACKED = false, bool MED = false, bool FEW = false>
__global__ __launch_bounds__(FTHREADS, OCC) void lrg_fused_stack_kernel(LrgFusedArgs args) {
    constexpr int FM = 32 * RT;      // rows (points) per workgroup
    static_assert(!PACKED || RT == 1, "packed rows use 32-row tiles");
    // FEW (the loop at a few dozen slots per lane: ~150 tiles per launch, a launch lasts as long as its slowest tile): accounted 256
    // VGPRs, so that at most two workgroups share a CU -- with three allowed the other lane's tiles double and triple up on
    // CUs while others idle.  585.7 k -> 596.6 k instance-steps/s at 68 rooms on two lanes (alternating runs, tools/r02_excl2.sh);
    // with hundreds of slots in flight (several tiles per CU wanted) it costs 8 %, hence the switch; 512 (a CU per tile) loses 5 %.
    if constexpr (PACKED && FEW) asm volatile("" ::: "v255");
    extern __shared__ __attribute__((aligned(16))) float smem[];
#if LRG_TRACE
    __shared__ long long lrg_trace_sh[32];
    if (threadIdx.x < 32) lrg_trace_sh[threadIdx.x] = 0;
#endif
    float *buf0 = smem;                       // outputs of even layers
    float *buf1 = smem + CAP0;                // the staged input and outputs of odd layers
    float *poolbuf = smem + CAP0 + CAP1;      // [512] running column maxima of the pooled layer / final-layer weights
    int *run_start = reinterpret_cast<int *>(poolbuf + 512);   // PACKED: [FM + 1] first row of each run (and the end)
    int *run_inst = run_start + FM + 1;                        //         [FM] instance of each run, -1 = dead rows past *nrows
    int *run_count = run_inst + FM;                            //         [1]
    int *row_run = run_count + 1;                              //         [FM] run of each row (medians in the launch)

    // PACKED: a one-dimensional grid with the problems (the two branches / the two heads) interleaved -- workgroup j is tile
    // j / nprob of problem j % nprob -- so that the live tiles are the FIRST workgroups of the launch: the dispatcher deals
    // consecutive workgroups round the XCDs and CUs, and <= 256 live tiles get a CU each.  With a (tiles, problem) grid the
    // second problem's tiles arrive after ~1000 dead workgroups and double up on CUs of the first's while others idle
    // (118 of 248 tiles shared a CU and took 75 k cycles instead of 51 k, profiles/r02_branch_tile_placement.txt).
    // The compacted tile lists of lrg_forward_rows are launched the same way.
    // With two problems of t0 and t1 live tiles the first 2 * min(t0, t1) workgroups alternate and the rest of the longer
    // problem follows, so the live tiles are exactly the first t0 + t1 workgroups.
    int bid = (int)blockIdx.x;
    if constexpr (PACKED && MED) {
        // the median workgroups come first in the grid (LrgFusedMedians); an instantiation of its own: their register keys
        // would cost the plain tile kernels their third and fourth workgroup per CU
        if (args.nmed > 0) {
            if (bid < args.nmed) {
                lrg_fused_median_wg(args.med, bid, reinterpret_cast<int *>(smem));
                return;
            }
            bid -= args.nmed;
        }
    }
    const int nprob = args.nprob;                 // > 0: interleaved
    int prob = nprob > 0 ? bid % nprob : (int)blockIdx.y;
    int bx = nprob > 0 ? bid / nprob : (int)blockIdx.x;
    if (nprob == 2) {
        auto live = [&](const LrgFusedProb &Q) {
            return PACKED ? (*Q.nrows + FM - 1) / FM : Q.tile_list ? *Q.tile_count : (int)(Q.rows / FM);
        };
        const int t0 = live(args.p[0]), t1 = live(args.p[1]);
        const int m = t0 < t1 ? t0 : t1;
        if (bid >= 2 * m) {
            prob = t0 > t1 ? 0 : 1;
            bx = bid - m;
            if (bx >= (t0 > t1 ? t0 : t1)) return;
        }
    }
    const LrgFusedProb &P = args.p[prob];
    // Tile-major block order (instance fastest): block b runs on XCD b % 8, and with duplicate-row skipping mostly the
    // FIRST tiles of the instances survive -- instance-major order would put all of them on one XCD.
    const int ninst = (int)(P.rows / P.rows_per_inst);
    int inst, tile, nvalid = 0x7fffffff;
    int nrows_packed = 0;
    if (PACKED) {
        nrows_packed = *P.nrows;
        if ((long)bx * FM >= nrows_packed) return;
        inst = 0;
        tile = bx;
    } else if (P.tile_list) {
        // a compacted list of the live tiles (lrg_prepare): workgroups 0 .. count-1 work, the rest leave at once -- the
        // dispatcher deals consecutive workgroups round the XCDs and CUs, so the live ones are spread evenly
        if (bx >= *P.tile_count) return;
        const int code = P.tile_list[bx];
        inst = code >> 6;
        tile = code & 63;
    } else {
        inst = bx % ninst;
        tile = bx / ninst;
        // rows beyond valid[instance] are copies of earlier rows (the padding rule, test_region_grow.py:240,:252):
        // their per-point results are identical and the max-pool ignores duplicates, so whole tiles of them are skipped
        // (the count is fetched here and tested after the input rows are staged: one memory round trip instead of two)
        if (P.valid) nvalid = P.valid[inst];
    }
    if (!PACKED && tile * FM >= P.rows_per_inst) return;
    const long r0 = PACKED ? (long)tile * FM : (long)inst * P.rows_per_inst + (long)tile * FM;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    TRACE(0);
#if LRG_TRACE
    if (CAP0 == LRG_TRACE && tid == 0) lrg_trace_sh[23] = (long long)wall_clock64();     // 100 MHz: calibrates the cycle counter
#endif

    // A 64-wide layer of a 64-row tile is laid out 2x2 (each wave one 32x32 tile) instead of 1x4 strips of which only
    // two would have columns: all four SIMDs stay busy through the narrow layers.
    auto is22 = [&](const LrgFusedLayer &L, int l) { return RT == 2 && L.N == 64 && (L.K == 64 || l == 0); };
    auto col_of = [&](const LrgFusedLayer &L, int l, int cb) { return is22(L, l) ? (wn & 1) * 32 : cb * FBN + wn * 32; };

    // this lane's float4 of k-group 0 of the 32-column block starting at column c of layer L (packed image)
    auto wptr = [&](const LrgFusedLayer &L, int c) {
        return reinterpret_cast<const float4 *>(L.w) + (long)(c >> 5) * L.ng * 64 + lane;
    };
    // bias of column c of layer L for this lane (a per-instance row when the layer carries the hoisted pooled product)
    auto bias_of = [&](const LrgFusedLayer &L, int c) -> float {
        if (!L.bias) return 0.f;
        if (PACKED && (L.flags & LRG_FL_INST_BIAS)) return 0.f;       // added per run of rows in the epilogue
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
    bool runs_done = false;
    if (PACKED && P.ctag) {
        // The centres come from this launch's median workgroups.  Runs first (whose centres does the tile need), the raw rows
        // requested meanwhile, then one coherent load per (run, centred channel) until its tag is this iteration's.
        constexpr int NE = (FM * 16 + FTHREADS - 1) / FTHREADS;
        float xv[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = tid + e * FTHREADS, row = idx / Kp, c = idx - row * Kp;
            xv[e] = (idx < FM * Kp && c < Kin && r0 + row < nrows_packed) ? P.x[(r0 + row) * P.ldx + c] : 0.f;
        }
        if (tid < 64) {
            const int row = tid & 31;
            const int mine = (r0 + row < nrows_packed) ? P.row_inst[r0 + row] : -1;
            const int prev = __shfl_up(mine, 1);
            const bool start = tid < 32 && (row == 0 || mine != prev);
            const unsigned long long m = __ballot(start);
            const int k = __popcll(m & ((2ull << row) - 1ull)) - 1;          // run of this row
            if (start) { run_start[k] = row; run_inst[k] = mine; }
            if (tid < 32) row_run[row] = k;
            if (tid == 0) { const int n = __popcll(m); run_start[n] = FM; *run_count = n; }
        }
        __syncthreads();
        for (int t = tid; t < *run_count * 16; t += FTHREADS) {
            const int k = t >> 4, c = t & 15;
            {
                const int ins = run_inst[k];
                float cv = 0.f;
                if (ins >= 0 && (P.cmask >> c & 1u)) {
                    const unsigned want = (unsigned)P.tags[2 * ins + 1];
                    unsigned long long v = 0;
                    for (int spin = 0; spin < (1 << 22); ++spin) {              // (bounded: a lost producer shows as wrong results, not a hang)
                        v = __hip_atomic_load(&P.ctag[ins * 16 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(v >> 32) == want) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    cv = __uint_as_float((unsigned)v);
                }
                poolbuf[t] = cv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = tid + e * FTHREADS, row = idx / Kp, c = idx - row * Kp;
            if (idx < FM * Kp)
                buf1[row * ld_x + c] = (c < Kin && r0 + row < nrows_packed) ? __fsub_rn(xv[e], poolbuf[row_run[row] * 16 + c]) : 0.f;
        }
        runs_done = true;
    } else if ((P.ldx & 3) == 0 && (Kin & 3) == 0 && (((uintptr_t)P.x) & 15) == 0 && !(PACKED && P.center)) {
        const int q = Kp >> 2;
        for (int idx = tid; idx < FM * q; idx += FTHREADS) {
            int row = idx / q, c4 = idx - row * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * c4 < Kin) v = *reinterpret_cast<const float4 *>(P.x + (r0 + row) * P.ldx + 4 * c4);
            *reinterpret_cast<float4 *>(&buf1[row * ld_x + 4 * c4]) = v;
        }
    } else if (PACKED && P.center) {
        // uncentred rows: subtract the owning instance's centre while staging (same float32 subtraction the gather would do)
        for (int idx = tid; idx < FM * Kp; idx += FTHREADS) {
            int row = idx / Kp, c = idx - row * Kp;
            float v = 0.f;
            if (c < Kin) {
                const int ins = (r0 + row < nrows_packed) ? P.row_inst[r0 + row] : -1;
                const float xv = P.x[(r0 + row) * P.ldx + c];
                v = ins >= 0 ? __fsub_rn(xv, P.center[ins * 16 + c]) : 0.f;
            }
            buf1[row * ld_x + c] = v;
        }
    } else {
        for (int idx = tid; idx < FM * Kp; idx += FTHREADS) {
            int row = idx / Kp, c = idx - row * Kp;
            buf1[row * ld_x + c] = c < Kin ? P.x[(r0 + row) * P.ldx + c] : 0.f;
        }
    }
    // poolbuf: running column maxima of a pooled stack, or the final [C,2] layer of a head (C <= 256)
    if (P.fw) { for (int i = tid; i < 2 * P.L[P.nlayers - 1].N; i += FTHREADS) poolbuf[i] = P.fw[i]; }
    else if (!runs_done) { for (int i = tid; i < 512; i += FTHREADS) poolbuf[i] = 0.f; }      // (PACKED keeps its maxima elsewhere)
    if (tile * FM >= nvalid) return;             // workgroup-uniform
    if (PACKED && !runs_done && tid < 64) {
        // runs of equal instance among the tile's rows (rows past *nrows: instance -1), found by wave 0 with one ballot
        const int row = tid & 31;
        const int mine = (r0 + row < nrows_packed) ? P.row_inst[r0 + row] : -1;
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
    __syncthreads();
    TRACE(1);
    const int nruns = PACKED ? *run_count : 1;

    const int nlayers = P.nlayers;
    int prevN = Kp;
    int lastN = 0, lastflags = 0;
    for (int l = 0; l < nlayers; ++l) {
        const LrgFusedLayer L = Lnext;             // descriptors are fetched one layer ahead (scalar loads off the critical path)
        if (l + 1 < nlayers) Lnext = P.L[l + 1];   // (after the last layer Lnext == L: the ring refill stays in bounds)
        const bool inplace = (L.flags & LRG_FL_INPLACE) != 0;
        const float *act_in = (l & 1) ? buf0 : buf1;
        float *act_out = ((l & 1) != 0) == !inplace ? buf1 : buf0;
        const int ld_in = prevN + 4, ld_out = L.N + 4;
        const bool m22 = is22(L, l);
        const int rbase = m22 ? (wn >> 1) * 32 : 0;          // first row of this wave's strip within the tile
        const int ntile = m22 ? 1 : RT;
        const float *ap = act_in + (rbase + li) * ld_in + 4 * lh;
        const int ncb = m22 ? 1 : (L.N + FBN - 1) / FBN;
        TRACE(2 + 2 * l);
        for (int cb = 0; cb < ncb; ++cb) {
            const int col0 = col_of(L, l, cb);
            const bool wave_on = col0 < L.N;         // a 64-wide layer outside the 2x2 layout keeps two of the four waves busy
            // the pass after this one: next column block, else the next layer's first
            const bool same = cb + 1 < ncb;
            const LrgFusedLayer &Lx = same ? L : Lnext;
            int coln = col_of(Lx, same ? l : l + 1, same ? cb + 1 : 0);
            if (coln >= Lx.N) coln = 0;
            const float4 *wpn = wptr(Lx, coln);

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
            if constexpr (PACKED) {
                if (wave_on && (L.flags & LRG_FL_INST_BIAS) && L.bias) {
#pragma unroll
                    for (int k = 0; k < RB; ++k) {
                        const int ins = k < nruns ? run_inst[k] : -1;
                        if (ins >= 0) bk[k] = L.bias[(long)ins * L.N + col0 + li];
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
                            tile_mfma_first<RT, 1>(acc, ap, ld_in, wp, L.ng, bf, cb == 0);
                        }
                    } else if (L.K == 128) tile_mfma<16, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else if (L.K == 64) tile_mfma<8, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else if (L.K == 256) tile_mfma<32, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else {
                        prefetch_b<FD>(bq, wpn);
                        tile_mfma_first<RT, RT>(acc, ap, ld_in, wp, L.ng, bf, cb == 0);
                    }
                } else {
                    if (L.K == 128) tile_mfma<16, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else if (L.K == 64) tile_mfma<8, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else if (L.K == 256) tile_mfma<32, RT, RT, FD>(acc, ap, ld_in, wp, wpn, bq);
                    else {
                        prefetch_b<FD>(bq, wpn);
                        tile_mfma_first<RT, RT>(acc, ap, ld_in, wp, L.ng, bf, cb == 0);
                    }
                }
            } else {
                prefetch_b<FD>(bq, wpn);      // an idle wave still owes the next pass its first weights
            }
            TRACE_PASS(l, cb, 0);
            if (inplace) __syncthreads();            // the output overlays this layer's input: everyone must be done reading
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
                                add_run(k, ins >= 0 ? L.bias[(long)ins * L.N + col] : 0.f);
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
                                    const int ins = run_inst[k];
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
        __syncthreads();                             // layer boundary: outputs visible, inputs dead
        if constexpr (PACKED) {
            if ((L.flags & LRG_FL_POOL) && !(L.flags & LRG_FL_KEEP)) {
                // the parked per-run maxima -> the pooled features (:122-125), coalesced, nothing waits for them
                const int runcap = (act_out == buf1 ? CAP1 : CAP0) / L.N;
                const int nk = nruns < runcap ? nruns : runcap;
                for (int k = 0; k < nk; ++k) {
                    const int ins = run_inst[k];
                    if (ins < 0) continue;
                    int *dst = reinterpret_cast<int *>(P.pool + (long)ins * P.pool_stride);
                    for (int c = tid; c < L.N; c += FTHREADS) {
                        const int m = reinterpret_cast<const int *>(act_out)[k * L.N + c];
                        if (m > 0) atomicMax(dst + c, m);
                    }
                }
            }
        }
        if (L.gout && (L.flags & LRG_FL_KEEP) && !inplace) {
            // HBM copy of a layer the next one reads from LDS (conv[1] for the heads, :130,:134): whole rows, float4
            const int q = L.N >> 2;
            float *gb = L.gout + r0 * L.N;
            for (int idx = tid; idx < FM * q; idx += FTHREADS) {
                const int row = idx / q, c4 = idx - row * q;
                *reinterpret_cast<float4 *>(gb + (unsigned)(row * L.N + 4 * c4)) =
                    *reinterpret_cast<const float4 *>(act_out + row * ld_out + 4 * c4);
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
        if (q == 0) *reinterpret_cast<float2 *>(P.fout + (r0 + row) * 2) = make_float2(s0 + P.fb[0], s1 + P.fb[1]);
    }
    // ---- leave the pooled feature of this instance zero for the next evaluation (it was consumed by the GEMV) ----
    if (!PACKED && P.zero_pool && tile == 0)
        for (int c = tid; c < P.zero_count; c += FTHREADS) P.zero_pool[(long)inst * P.zero_count + c] = 0.f;
    TRACE(20);
#if LRG_TRACE
    if (CAP0 == LRG_TRACE && tid == 0 && g_lrg_trace && bx < 2048) {
        lrg_trace_sh[21] = nruns;
        lrg_trace_sh[24] = (long long)wall_clock64();
        // where the workgroup ran: HW_ID (wave / simd / cu / sh / se) and XCC_ID
        lrg_trace_sh[22] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        for (int i = 0; i < 32; ++i) g_lrg_trace[((long)prob * 2048 + bx) * 32 + i] = lrg_trace_sh[i];
    }
#endif
}

template <int CAP0, int CAP1, int RT, int FD, int OCC, bool DIRECT, bool PACKED = false, bool MED = false, bool FEW = false>
static int launch_stack(const LrgFusedArgs &a, int nprob, hipStream_t st) {
    constexpr int FM = 32 * RT;
    long maxrows = 0;
    for (int i = 0; i < nprob; ++i) {
        const LrgFusedProb &P = a.p[i];
        if (P.rows % FM != 0 || (!PACKED && P.rows_per_inst % FM != 0)) return LRG_EINVAL - 30;
        if (PACKED != (P.nrows != nullptr) || (PACKED && !P.row_inst)) return LRG_EINVAL - 30;
        if (P.nlayers < 1 || P.nlayers > LRG_FUSED_MAXL) return LRG_EINVAL - 31;
        const int Kp = (P.Kin + 7) & ~7;
        if (FM * (Kp + 4) > CAP1) return LRG_EINVAL - 32;
        for (int l = 0; l < P.nlayers; ++l) {
            const LrgFusedLayer &L = P.L[l];
            if (L.N % 64 != 0 || L.N > 512 || L.ng != (L.K + 7) / 8) return LRG_EINVAL - 33;
            if (l > 0 && (L.K != P.L[l - 1].N || (L.K != 64 && L.K != 128 && L.K != 256))) return LRG_EINVAL - 34;
            if (l == 0 && (L.K != P.Kin || (L.K != 64 && L.K != 128 && L.K != 256 && L.K > 56))) return LRG_EINVAL - 35;
            const bool inplace = (L.flags & LRG_FL_INPLACE) != 0;
            const bool to_buf1 = ((l & 1) != 0) == !inplace;
            if ((L.flags & LRG_FL_KEEP) && FM * (L.N + 4) > (to_buf1 ? CAP1 : CAP0)) return LRG_EINVAL - 36;
            if (inplace && (RT != 1 || L.N > FBN || l + 1 != P.nlayers)) return LRG_EINVAL - 38;   // single column block, last layer only
            if (l + 1 < P.nlayers && !(L.flags & LRG_FL_KEEP)) return LRG_EINVAL - 37;
        }
        if (P.fw && (P.L[P.nlayers - 1].N > 256 || (P.L[P.nlayers - 1].flags & LRG_FL_POOL))) return LRG_EINVAL - 39;   // share the 512-float scratch
        if (P.rows > maxrows) maxrows = P.rows;
    }
    if (maxrows == 0) return 0;
    const size_t lds = (size_t)(CAP0 + CAP1 + 512 + (PACKED ? 3 * FM + 8 : 0)) * sizeof(float);
    auto kern = lrg_fused_stack_kernel<CAP0, CAP1, RT, FD, OCC, DIRECT, PACKED, MED, FEW>;
    static bool attr_done[LRG_MAX_DEVICES] = {};      // per instantiation, per device
    const int dev = lrg_current_device();
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
        attr_done[dev] = true;
    }
    bool lists = true;
    for (int i = 0; i < nprob; ++i) lists = lists && a.p[i].tile_list != nullptr;
    if (PACKED || lists) {
        LrgFusedArgs b = a;
        b.nprob = nprob;
        if (!MED) b.nmed = 0;
        hipLaunchKernelGGL(kern, dim3((unsigned)(maxrows / FM) * nprob + (unsigned)b.nmed), dim3(FTHREADS), lds, st, b);
    } else {
        LrgFusedArgs b = a;
        b.nprob = 0;
        b.nmed = 0;
        hipLaunchKernelGGL(kern, dim3((unsigned)(maxrows / FM), nprob), dim3(FTHREADS), lds, st, b);
    }
    LRG_LAUNCH_CHECK();
    return 0;
}

static bool needs_direct(const LrgFusedArgs &a, int nprob) {
    for (int i = 0; i < nprob; ++i)
        for (int l = 0; l < a.p[i].nlayers; ++l) {
            const LrgFusedLayer &L = a.p[i].L[l];
            if (L.gout && (!(L.flags & LRG_FL_KEEP) || (L.flags & LRG_FL_INPLACE))) return true;
        }
    return false;
}

// Three workgroups (one wave per SIMD each) per CU: 53 KB / 44 KB of LDS and <= 168 VGPRs.  While one workgroup is in
// its barrier-separated narrow layers or staging its rows, the other two keep the MFMA pipe busy.
int lrg_fused_branches(const LrgFusedArgs &a, int nprob, hipStream_t st) {
    // lite 0/1/2: hidden widths 64 / 128
    if (needs_direct(a, nprob)) return launch_stack<64 * 68, 64 * 132, 2, 4, 2, true>(a, nprob, st);
    // With per-instance row counts (the grow loop: most sets are far below 512 distinct rows) 32-row tiles waste fewer
    // padded rows than 64-row ones (+7 % loop throughput); dense batches keep the 64-row tile, whose two stacked MFMA
    // tiles share every weight operand.
    if (a.p[0].valid) return launch_stack<32 * 68, 32 * 132, 1, 4, 4, false>(a, nprob, st);    // (5 per CU spills: slower)
    return launch_stack<64 * 68, 64 * 132, 2, 4, 3, false>(a, nprob, st);
}

int lrg_fused_heads(const LrgFusedArgs &a, int nprob, hipStream_t st) {
    // 64 -> 256 -> 128 (-> 2); the last hidden layer is written in place
    if (needs_direct(a, nprob)) return launch_stack<32 * 260, 32 * 68, 1, 4, 2, true>(a, nprob, st);
    return launch_stack<32 * 260, 32 * 68, 1, 4, 3, false>(a, nprob, st);
}

// Build-time knobs of the packed-row kernels (tools/r02_fd2.sh): depth of the weight ring and workgroups per CU.  Measured on
// the loop (profiles/r02_weight_ring_depth_experiment*.txt): 4 / 8 k-groups in flight give the same kernel durations both at
// ~2 tiles per CU and at < 1 tile per CU -- the lone tile is bound by its per-pass epilogues and layer barriers, not by weights.
#ifndef LRG_PACKED_FD
#define LRG_PACKED_FD 4
#endif
#ifndef LRG_PACKED_OCC
#define LRG_PACKED_OCC 3
#endif
#ifndef LRG_PACKED_HEAD_FD
#define LRG_PACKED_HEAD_FD 4
#endif
int lrg_fused_median_workgroups(int n_slots) { return n_slots * LRG_MED_SPLIT; }

int lrg_fused_branches_packed(const LrgFusedArgs &a, int nprob, hipStream_t st) {
    // lite 1: conv[1] is the pooled layer itself -- it does not stay in LDS, so its HBM copy (read by the heads) is stored
    // from the accumulators
    if (a.nmed > 0) {
        if (needs_direct(a, nprob)) return launch_stack<32 * 68, 32 * 132, 1, 4, 2, true, true, true>(a, nprob, st);
        return launch_stack<32 * 68, 32 * 132, 1, LRG_PACKED_FD, 2, false, true, true>(a, nprob, st);
    }
    if (needs_direct(a, nprob)) return launch_stack<32 * 68, 32 * 132, 1, 4, 2, true, true>(a, nprob, st);
    if (a.few) return launch_stack<32 * 68, 32 * 132, 1, LRG_PACKED_FD, LRG_PACKED_OCC, false, true, false, true>(a, nprob, st);
    return launch_stack<32 * 68, 32 * 132, 1, LRG_PACKED_FD, LRG_PACKED_OCC, false, true>(a, nprob, st);
}

int lrg_fused_heads_packed(const LrgFusedArgs &a, int nprob, hipStream_t st) {
    if (a.few && !needs_direct(a, nprob)) return launch_stack<32 * 260, 32 * 68, 1, LRG_PACKED_HEAD_FD, 3, false, true, false, true>(a, nprob, st);
    return launch_stack<32 * 260, 32 * 68, 1, LRG_PACKED_HEAD_FD, 3, false, true>(a, nprob, st);
}
