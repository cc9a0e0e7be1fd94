// Fused LrgNet stacks for gfx950: a whole branch (learn_region_grow_util.py:106-123) or a whole head (:138-162)
// per 64-row (branch) / 32-row (head) tile in ONE kernel.  Activations never leave the CU: they ping-pong between two
// LDS buffers; the layer weights (L2-resident, 3.2 MB in all, pre-arranged in operand order by lrg_pack_weights) go
// straight into the MFMA B operands through a register ring that runs a few k-groups ahead of the fp32 MFMAs
// (v_mfma_f32_32x32x2_f32, exact fp32) and never drains between layers.  Only conv[1] (needed by the heads), the
// pooled maxima and the logits are written to HBM; the 512-wide layer and the 1088-wide concat are never materialised.
// The tile itself is lrg_fused_tile (lrg_fused_tile.inl), shared with the free-running region-grow kernel (lrg_async.inl).
//
// The layer-streamed formulation (one launch per layer, lrg_net.hip) stays available: it is what the layer-by-layer
// parity tests and the "HBM-streamed" roofline figure use.
#include "lrg_common.h"
#include "lrg_fused.h"
#include "lrg_fused_tile.inl"
#include "lrg_stream_layer.inl"

#if LRG_TRACE
__device__ long long *g_lrg_trace = nullptr;
extern "C" void lrg_set_trace(long long *p) { hipMemcpyToSymbol(HIP_SYMBOL(g_lrg_trace), &p, sizeof(p)); }
#endif

template <int CAP0, int CAP1, int RT, int FD, int OCC, bool DIRECT, bool PACKED = false, bool FEW = false>
__global__ __launch_bounds__(FTHREADS, OCC) void lrg_fused_stack_kernel(LrgFusedArgs args) {
    constexpr int FM = 32 * RT;      // rows (points) per workgroup
    // FEW (the loop at a few dozen slots per lane: ~150 tiles per launch, a launch lasts as long as its slowest tile): accounted 256
    // VGPRs, so that at most two workgroups share a CU -- with three allowed the other lane's tiles double and triple up on
    // CUs while others idle.  585.7 k -> 596.6 k instance-steps/s at 68 rooms on two lanes (alternating runs, tools/r02_excl2.sh);
    // with hundreds of slots in flight (several tiles per CU wanted) it costs 8 %, hence the switch; 512 (a CU per tile) loses 5 %.
    if constexpr (PACKED && FEW) asm volatile("" ::: "v255");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    long long *trace_sh = nullptr;
#if LRG_TRACE
    __shared__ long long lrg_trace_sh[32];
    if (threadIdx.x < 32) lrg_trace_sh[threadIdx.x] = 0;
    trace_sh = lrg_trace_sh;
#endif
    // PACKED: a one-dimensional grid with the problems (the two branches / the two heads) interleaved -- workgroup j is tile
    // j / nprob of problem j % nprob -- so that the live tiles are the FIRST workgroups of the launch: the dispatcher deals
    // consecutive workgroups round the XCDs and CUs, and <= 256 live tiles get a CU each.  With a (tiles, problem) grid the
    // second problem's tiles arrive after ~1000 dead workgroups and double up on CUs of the first's while others idle
    // (118 of 248 tiles shared a CU and took 75 k cycles instead of 51 k, profiles/r02_branch_tile_placement.txt).
    // The compacted tile lists of lrg_forward_rows are launched the same way.
    // With two problems of t0 and t1 live tiles the first 2 * min(t0, t1) workgroups alternate and the rest of the longer
    // problem follows, so the live tiles are exactly the first t0 + t1 workgroups.
    const int bid = (int)blockIdx.x;
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
#if LRG_TRACE
    if (CAP0 == LRG_TRACE && threadIdx.x == 0) lrg_trace_sh[23] = (long long)wall_clock64();     // 100 MHz: calibrates the cycle counter
#endif
    const int nruns = lrg_fused_tile<CAP0, CAP1, RT, FD, DIRECT, PACKED, false>(P, r0, inst, tile, nvalid, nrows_packed, smem, LrgWgTeam(), trace_sh);
    (void)nruns;
#if LRG_TRACE
    if (CAP0 == LRG_TRACE && threadIdx.x == 0 && g_lrg_trace && bx < 2048) {
        lrg_trace_sh[21] = nruns;
        lrg_trace_sh[24] = (long long)wall_clock64();
        // where the workgroup ran: HW_ID (wave / simd / cu / sh / se) and XCC_ID
        lrg_trace_sh[22] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        for (int i = 0; i < 32; ++i) g_lrg_trace[((long)prob * 2048 + bx) * 32 + i] = lrg_trace_sh[i];
    }
#endif
}

template <int CAP0, int CAP1, int RT, int FD, int OCC, bool DIRECT, bool PACKED = false, bool FEW = false>
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
    const size_t lds = (size_t)LRG_TILE_LDS_FLOATS(CAP0, CAP1, RT, PACKED) * sizeof(float);
    auto kern = lrg_fused_stack_kernel<CAP0, CAP1, RT, FD, OCC, DIRECT, PACKED, FEW>;
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
        hipLaunchKernelGGL(kern, dim3((unsigned)(maxrows / FM) * nprob), dim3(FTHREADS), lds, st, b);
    } else {
        LrgFusedArgs b = a;
        b.nprob = 0;
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

// ONE layer per launch (lrg_forward's layer-streamed formulation, LRG_FWD_STREAM_TILES).  LrgNet's own layer shapes run on the streaming wavefront kernel
// (lrg_stream_layer.inl: both MFMA operands from memory in operand order, no LDS, no barrier); any other shape as a 1-layer stack on the fused tile, output from
// the accumulators to HBM.  Either way the MFMA sequence and the epilogue of the fused stacks: the same bits.
// LRG_LAYER_VARIANT (measurements, profiles/r05_layer_variants.txt): 0 / 2 = the fused tile for every shape (64-row, two per CU / 32-row, four per CU),
// 3 = the streaming kernel with its wider column groups, 4 (default) = with the narrower ones.
int lrg_fused_layer(const LrgFusedArgs &a, int nprob, hipStream_t st) {
    int K = 0;
    for (int i = 0; i < nprob; ++i) {
        if (a.p[i].nlayers != 1 || !a.p[i].L[0].gout || (a.p[i].L[0].flags & (LRG_FL_KEEP | LRG_FL_INPLACE | LRG_FL_POOL)) || a.p[i].fw) return LRG_EINVAL - 30;
        K = a.p[i].Kin > K ? a.p[i].Kin : K;
    }
    static const int v = getenv("LRG_LAYER_VARIANT") ? atoi(getenv("LRG_LAYER_VARIANT")) : 5;
    if (v >= 3) {
        const int ng = a.p[0].L[0].ng, N = a.p[0].L[0].N;
        bool same = true, aligned = true;
        for (int i = 0; i < nprob; ++i) {
            same = same && a.p[i].L[0].ng == ng && a.p[i].L[0].N == N && a.p[i].Kin == a.p[0].Kin;
            same = same && a.p[i].L[0].bias != nullptr;
            aligned = aligned && a.p[i].ldx % 4 == 0 && a.p[i].Kin == 8 * ng && a.p[i].rows_per_inst % 32 == 0;
        }
        if (v >= 5) {      // the persistent form, the weights from LDS (7: every ring eight k-groups deep, refilled a line at a time)
            if (same && aligned) {
                if (ng == 8 && N == 64) return lrg_stream_layer_lds_launch<8, 2, 8, 4, false>(a, nprob, st);
                if (ng == 8 && N == 128) return v == 7 ? lrg_stream_layer_lds_launch<8, 4, 8, 3, false>(a, nprob, st) : lrg_stream_layer_lds_launch<8, 4, 4, 3, false>(a, nprob, st);
                if (ng == 8 && N == 256) return v == 7 ? lrg_stream_layer_lds_launch<8, 4, 8, 3, false>(a, nprob, st) : lrg_stream_layer_lds_launch<8, 4, 4, 3, false>(a, nprob, st);
                if (ng == 16 && N == 512) return v == 7 ? lrg_stream_layer_lds_launch<16, 4, 8, 2, false>(a, nprob, st) : lrg_stream_layer_lds_launch<16, 4, 4, 2, false>(a, nprob, st);
                if (ng == 32 && N == 128) return v == 7 ? lrg_stream_layer_lds_launch<32, 2, 8, 2, false>(a, nprob, st) : lrg_stream_layer_lds_launch<32, 2, 4, 2, false>(a, nprob, st);
            }
        }
        if (same && ng == 2 && N == 64 && a.p[0].rows_per_inst % 32 == 0) return lrg_stream_layer_launch<2, 2, 2, 4, true>(a, nprob, st);
        if (same && aligned) {
            if (ng == 8 && N == 64) return lrg_stream_layer_launch<8, 2, 4, 4, false>(a, nprob, st);
            if (ng == 8 && N == 128) return v == 3 ? lrg_stream_layer_launch<8, 4, 4, 3, false>(a, nprob, st) : lrg_stream_layer_launch<8, 2, 4, 4, false>(a, nprob, st);
            if (ng == 8 && N == 256) return v == 3 ? lrg_stream_layer_launch<8, 8, 3, 2, false>(a, nprob, st) : lrg_stream_layer_launch<8, 4, 4, 3, false>(a, nprob, st);
            if (ng == 16 && N == 512) return v == 3 ? lrg_stream_layer_launch<16, 8, 3, 2, false>(a, nprob, st) : lrg_stream_layer_launch<16, 4, 4, 3, false>(a, nprob, st);
            if (ng == 32 && N == 128) return v == 3 ? lrg_stream_layer_launch<32, 4, 4, 3, false>(a, nprob, st) : lrg_stream_layer_launch<32, 2, 4, 4, false>(a, nprob, st);
        }
    }
    // no layer output stays in LDS, so the even buffer is a stub and four 32-row tiles share a CU (one tile's load / MFMA / store phases do not overlap; the
    // neighbours' do; two 64-row tiles per CU: 4.78 ms per evaluation of 1088 instances against 4.59)
    if (v == 0) {
        if (K <= 128) return launch_stack<64 * 68, 64 * 132, 2, 4, 2, true>(a, nprob, st);      // (the instantiation of the keep-everything parity path)
        return launch_stack<32 * 68, 32 * 260, 1, 4, 3, true>(a, nprob, st);                    // K = 256: the heads' second layer
    }
    if (K <= 64) return launch_stack<64, 32 * 68, 1, 4, 4, true>(a, nprob, st);
    if (K <= 128) return launch_stack<64, 32 * 132, 1, 4, 4, true>(a, nprob, st);
    return launch_stack<64, 32 * 260, 1, 4, 4, true>(a, nprob, st);
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
int lrg_fused_branches_packed(const LrgFusedArgs &a, int nprob, hipStream_t st) {
    // lite 1: conv[1] is the pooled layer itself -- it does not stay in LDS, so its HBM copy (read by the heads) is stored
    // from the accumulators
    if (needs_direct(a, nprob)) return launch_stack<32 * 68, 32 * 132, 1, 4, 2, true, true>(a, nprob, st);
    if (a.few) return launch_stack<32 * 68, 32 * 132, 1, LRG_PACKED_FD, LRG_PACKED_OCC, false, true, true>(a, nprob, st);
    return launch_stack<32 * 68, 32 * 132, 1, LRG_PACKED_FD, LRG_PACKED_OCC, false, true>(a, nprob, st);
}

int lrg_fused_heads_packed(const LrgFusedArgs &a, int nprob, hipStream_t st) {
    if (a.few && !needs_direct(a, nprob)) return launch_stack<32 * 260, 32 * 68, 1, LRG_PACKED_HEAD_FD, 3, false, true, true>(a, nprob, st);
    return launch_stack<32 * 260, 32 * 68, 1, LRG_PACKED_HEAD_FD, 3, false, true>(a, nprob, st);
}
