// ONE 1x1-convolution layer of LrgNet over all rows, streamed (learn_region_grow_util.py:109-111, :138-141, :151-154 as lrg_forward runs them layer by layer with
// LRG_FWD_STREAM_TILES): y[r, :] = relu(x[r, :K] @ W + bias), rows in and out through HBM exactly once.
//
// A wavefront, not a workgroup, is the unit here: it owns 32 rows x (32 CT) columns, takes BOTH MFMA operands straight from memory in operand order -- no LDS, no
// barrier, nothing shared with the neighbouring wavefronts of its workgroup but the L1 lines they happen to have in common:
//   A: lane (row i = lane % 32, k half h = lane / 32) reads the float4 x[r0 + i, 8 g + 4 h ..] of k-group g: the four A values of that group's four
//      v_mfma_f32_32x32x2_f32 (the pairing of the fused tile and of the weight image: bit-identical sums).  A wave's request touches 32 rows x 32 bytes; the
//      eight requests of a 64-channel row use every byte of its two lines, three out of four from L1.
//   B: the lane's float4 of lrg_pack_weights' image, one per k-group and 32-column block (L2-resident).
// Both run D k-groups ahead of the matrix cores in a register ring; OCC wavefronts per SIMD cover each other's prologues (first operands on their way) and
// epilogues (bias, ReLU, 16 CT stores of two 128-byte row segments each).  The column groups of a row tile are neighbouring wavefronts of ONE workgroup: the
// tile's rows come from HBM once and from L1 for the others.
#pragma once
#include "lrg_fused.h"

typedef float lrg_sf32x16 __attribute__((ext_vector_type(16)));

// (the launch's own small argument block: every field is selected with the wave's problem bit -- scalar loads and selects; the stacks' LrgFusedArgs indexed by
//  a run-time problem number was read with vector loads, a trip to memory in front of the first operand request)
struct LrgStreamArgs {
    const float *x[2], *w[2], *bias[2];
    float *y[2];
    long rows[2];
    int ldx[2], rows_per_inst[2], inst_bias[2];
    int K, N, relu, nprob, cg_shift;
    int dbg;             // LRG_STREAM_DBG (experiments): 1 = no stores, 2 = the ring is not refilled, 4 = no B reads
};

template <int NG, int CT, int D, int OCC, bool FIRST>
__global__ __launch_bounds__(256, OCC) void lrg_stream_layer_kernel(LrgStreamArgs a) {
    static_assert(D >= 1 && D <= NG, "the ring is not deeper than the layer");
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // the wave's item: problems back to back, row tiles in order, column groups fastest
    const int N = a.N;
    unsigned item = blockIdx.x * 4u + (unsigned)wave;
    const unsigned n0 = (unsigned)(a.rows[0] >> 5) << a.cg_shift;       // (column groups per row tile: a power of two)
    const bool second = item >= n0;
    if (second) item -= n0;
    if (second && a.nprob < 2) return;
    const unsigned rt = item >> a.cg_shift;
    const int cg = (int)(item - (rt << a.cg_shift));
    const long r0 = (long)rt * 32;
    if (r0 >= (second ? a.rows[1] : a.rows[0])) return;
    const int col0 = cg * 32 * CT;
    const int ldx = second ? a.ldx[1] : a.ldx[0];

    // Buffer addressing: a uniform base (descriptor), ONE per-lane offset register per stream, the per-request part as the instruction's scalar offset -- with
    // flat pointers every request of the ring held an address pair of its own and the ring spilled.
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>((second ? a.w[1] : a.w[0]) + (size_t)(cg * CT) * NG * 256), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>((second ? a.x[1] : a.x[0]) + (size_t)r0 * ldx), 0, 0x7fffffff, 0x00020000);
    const unsigned vw = (unsigned)lane * 16u;                        // + ((c NG + g) 64) float4
    const unsigned vx = (unsigned)(li * ldx + 4 * lh) * 4u;          // + 8 g floats
    float4 ar[D], br[D][CT];
    auto request = [&](int g, int slot) {
        if constexpr (FIRST) {
            // a narrow input row (13 features, rows not 16-byte aligned): scalar loads, the k past K as zeros (the image's rows past K are zeros too)
            const int k0 = 8 * g + 4 * lh, K = a.K;
            const float q0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, 32 * g + 0, 0)), q1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, 32 * g + 4, 0)),
                        q2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, 32 * g + 8, 0)), q3 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, 32 * g + 12, 0));
            ar[slot].x = k0 + 0 < K ? q0 : 0.f;
            ar[slot].y = k0 + 1 < K ? q1 : 0.f;
            ar[slot].z = k0 + 2 < K ? q2 : 0.f;
            ar[slot].w = k0 + 3 < K ? q3 : 0.f;
        } else {
            const lrg_u32x4v u = __builtin_amdgcn_raw_buffer_load_b128(rx, vx, 32 * g, 0);
            ar[slot] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const lrg_u32x4v u = __builtin_amdgcn_raw_buffer_load_b128(rw, vw, (c * NG + g) * 1024, 0);
            br[slot][c] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        }
    };
#pragma unroll
    for (int g = 0; g < D; ++g) request(g, g);
    __builtin_amdgcn_sched_barrier(0);
    // bias of the lane's column in every block (a per-instance row for a head's first layer: the hoisted pooled product, :128-141)
    float bv[CT];
    {
        // (branch-free: with a test and a branch here the compiler moved the ring's first requests below them)
        const unsigned inst = (rt * 32u) / (unsigned)(second ? a.rows_per_inst[1] : a.rows_per_inst[0]);
        const float *b = (second ? a.bias[1] : a.bias[0]) + (size_t)((second ? a.inst_bias[1] : a.inst_bias[0]) ? inst : 0u) * (unsigned)N + col0;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(b), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int c = 0; c < CT; ++c) bv[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, (unsigned)li * 4u, 128 * c, 0));
    }
    lrg_sf32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int s = g % D;
        const float4 av = ar[s];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, br[s][c].x, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, br[s][c].y, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, br[s][c].z, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, br[s][c].w, acc[c], 0, 0, 0);
        if (g + D < NG) request(g + D, s);
        // (nothing moves across a k-group: left to itself the compiler sinks the ring's requests to their uses -- a trip to memory in front of every group)
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: bias, ReLU, out (lane (i, h) holds rows 4 h + (j & 3) + 8 (j >> 2) of column i of every block) ----
    const bool relu = a.relu != 0;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((second ? a.y[1] : a.y[0]) + (size_t)r0 * N + col0, 0, 0x7fffffff, 0x00020000);
    const unsigned vy = (unsigned)(4 * lh * N + li) * 4u;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = acc[c][j] + bv[c];
            if (relu) v = fmaxf(v, 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, vy, (((j & 3) + 8 * (j >> 2)) * N + 32 * c) * 4, 0);
        }
}

// ---- the persistent form: the weights from LDS ----
// With every wavefront taking its B operands from L2 a 32-row tile re-reads the layer's whole kernel: 256 KB for 80 KB of rows in the 128 -> 512 layer, ~6.3 TB/s of
// L2 reads for both MFMA-bound layers at 64-67 % of the matrix cores' peak (profiles/r05_layer_variants.txt).  Here a workgroup is bound to ONE column group for its
// lifetime: it stages that group's panel of the image in LDS once ([CT NG][64 lanes] float4, <= 64 KB) and its wavefronts (four, or eight where the panel is large: two such workgroups fill a CU) then walk over row tiles, each on
// its own -- A through the register ring from HBM (requests run on into the NEXT tile: no prologue bubble between tiles), B by ds_read_b128 one k-group ahead, no
// barrier after the first.  The column groups of a row tile are workgroups of the same XCD at the same place in their walks: the tile's rows come from HBM once
// and from that XCD's L2 for the others.
// RT = 2: a wavefront takes TWO consecutive row tiles at a time -- every B value read from LDS feeds two MFMAs.
template <int NG, int CT, int D, int OCC, bool FIRST, int W, int RT>
__global__ __launch_bounds__(64 * W, OCC) void lrg_stream_layer_lds_kernel(LrgStreamArgs a) {
    static_assert(D >= 1 && D <= NG && NG % D == 0, "ring slots are static: the ring's depth divides the layer's k-groups");
    extern __shared__ __attribute__((aligned(16))) float lrg_stream_smem[];
    float4 *bl = reinterpret_cast<float4 *>(lrg_stream_smem);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N;
    // workgroup -> (XCD, unit = problem x column group, place in the unit's walk)
    const unsigned units = (unsigned)a.nprob << a.cg_shift;
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const unsigned unit = j % units, idx = (j / units) * 8u + xcd, nidx = (gridDim.x >> 3) / units * 8u;
    const bool second = (unit >> a.cg_shift) != 0;
    const int cg = (int)(unit & ((1u << a.cg_shift) - 1u));
    const int col0 = cg * 32 * CT;
    const int ldx = second ? a.ldx[1] : a.ldx[0];
    const unsigned ntiles = (unsigned)((second ? a.rows[1] : a.rows[0]) >> 5) / (unsigned)RT;      // (units of RT row tiles)
    {
        const float4 *src = reinterpret_cast<const float4 *>(second ? a.w[1] : a.w[0]) + (size_t)(cg * CT) * NG * 64;
        for (int i = tid; i < CT * NG * 64; i += 64 * W) bl[i] = src[i];
    }
    __syncthreads();
    unsigned t = idx * (unsigned)W + (unsigned)wave;
    const unsigned tstep = nidx * (unsigned)W;
    if (t >= ntiles) return;
    const float *xbase = second ? a.x[1] : a.x[0];
    float *ybase = (second ? a.y[1] : a.y[0]) + col0;
    const float *bbase = (second ? a.bias[1] : a.bias[0]) + col0;
    const unsigned rpi = (unsigned)(second ? a.rows_per_inst[1] : a.rows_per_inst[0]);
    const bool inst_bias = (second ? a.inst_bias[1] : a.inst_bias[0]) != 0;
    const bool relu = a.relu != 0;
    const unsigned vx = (unsigned)(li * ldx + 4 * lh) * 4u;          // + 8 g floats
    const unsigned vx1 = vx + 32u * (unsigned)ldx * 4u;              // (the second tile of a pair)
    const unsigned vy = (unsigned)(4 * lh * N + li) * 4u;
    const unsigned vy1 = vy + 32u * (unsigned)N * 4u;
    const float4 *bp = bl + lane;                                    // + (c NG + g) 64
    float4 ar[D][RT];
    auto request = [&](const __amdgpu_buffer_rsrc_t &rx, int g, int slot) {
        if constexpr (FIRST) {
            static_assert(!FIRST || RT == 1, "narrow rows: one tile at a time");
            const int k0 = 8 * g + 4 * lh, K = a.K;
            const float q0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, 32 * g + 0, 0)), q1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, 32 * g + 4, 0)),
                        q2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, 32 * g + 8, 0)), q3 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, 32 * g + 12, 0));
            ar[slot][0].x = k0 + 0 < K ? q0 : 0.f;
            ar[slot][0].y = k0 + 1 < K ? q1 : 0.f;
            ar[slot][0].z = k0 + 2 < K ? q2 : 0.f;
            ar[slot][0].w = k0 + 3 < K ? q3 : 0.f;
        } else {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const lrg_u32x4v u = __builtin_amdgcn_raw_buffer_load_b128(rx, r ? vx1 : vx, 32 * g, 0);
                ar[slot][r] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
            }
        }
    };
    auto rsrc_of = [&](unsigned tile) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xbase + (size_t)tile * (32u * RT) * (unsigned)ldx), 0, 0x7fffffff, 0x00020000); };
    {
        const __amdgpu_buffer_rsrc_t rx = rsrc_of(t);
#pragma unroll
        for (int g = 0; g < D; ++g) request(rx, g, g);
    }
    for (;;) {
        const unsigned tn = t + tstep < ntiles ? t + tstep : t;       // (the last tile requests its own first groups again: nobody uses them)
        const __amdgpu_buffer_rsrc_t rx = rsrc_of(t), rxn = rsrc_of(tn);
        // bias of the lane's column in every block (a per-instance row for a head's first layer: the hoisted pooled product, :128-141)
        float bv[CT];
        {
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bbase + (size_t)(inst_bias ? (t * (32u * RT)) / rpi : 0u) * (unsigned)N), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int c = 0; c < CT; ++c) bv[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, (unsigned)li * 4u, 128 * c, 0));
        }
        lrg_sf32x16 acc[RT][CT];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;
        float4 b[2][CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) b[0][c] = bp[(c * NG) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int s = g % D;
            if (g + 1 < NG && !(a.dbg & 4))
#pragma unroll
                for (int c = 0; c < CT; ++c) b[(g + 1) & 1][c] = bp[(c * NG + g + 1) * 64];
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[s][r].x, b[g & 1][c].x, acc[r][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[s][r].y, b[g & 1][c].y, acc[r][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[s][r].z, b[g & 1][c].z, acc[r][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[s][r].w, b[g & 1][c].w, acc[r][c], 0, 0, 0);
            // The ring is refilled a 128-byte LINE of every row at a time (four k-groups = 4 x 32 bytes): the four requests go out back to back and the line is
            // fetched once.  One request per k-group touched each line four times a thousand cycles apart -- by then the CU's other wavefronts had pushed it
            // out of the 32 KB L1: four trips to L2 per line, as much L2 traffic as the weights were before they moved to LDS.
            constexpr bool LINES = !FIRST && D == 8 && NG % 4 == 0;
            if (a.dbg & 2) {
            } else if constexpr (LINES) {
                if ((g & 3) == 3)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int q = g + 5 + i;
                        if (q < NG) request(rx, q, q % D);
                        else request(rxn, q - NG, q % D);
                    }
            } else {
                if (g + D < NG) request(rx, g + D, s);
                else request(rxn, g + D - NG, s);
            }
            // the order inside a k-group: the NEXT group's B reads first (an LDS round trip ahead of their MFMAs, not right in front of them), the MFMAs, the ring's request
            if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, CT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * CT * RT, 0);
            if constexpr (LINES) {
                if ((g & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x020, 4 * RT, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x020, FIRST ? 4 : RT, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: bias, ReLU, out (lane (i, h) holds rows 4 h + (j & 3) + 8 (j >> 2) of column i of every block): 16 dword stores per block, two whole
        // 128-byte row segments each.  (Measured and dropped, profiles/r05_layer_variants.txt: the block through a wave-private LDS patch and out as 16-byte stores
        // of eight row segments per instruction -- 4 stores per block instead of 16 -- is 2-3 % slower; the tile computed transposed, 16-byte stores of 32-byte
        // pieces of 32 rows, 20-40 % slower.)
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(ybase + (size_t)t * (32u * RT) * (unsigned)N, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    float v = acc[r][c][jj] + bv[c];
                    if (relu) v = fmaxf(v, 0.f);
                    if (!(a.dbg & 1)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, r ? vy1 : vy, (((jj & 3) + 8 * (jj >> 2)) * N + 32 * c) * 4, 0);
                }
        if (tn == t) break;
        t = tn;
    }
}

template <int NG, int CT, int D, int OCC, bool FIRST, int W = 4, int RT = 1>
static int lrg_stream_layer_lds_launch(const LrgFusedArgs &a, int nprob, hipStream_t st) {
    LrgStreamArgs b = {};
    for (int i = 0; i < nprob; ++i) {
        const LrgFusedProb &P = a.p[i];
        const LrgFusedLayer &L = P.L[0];
        if (P.rows % (32 * RT) != 0 || P.rows_per_inst % (32 * RT) != 0 || L.N % (32 * CT) != 0 || L.ng != NG || !L.gout || !L.w || !L.bias) return LRG_EINVAL - 30;
        if (!FIRST && (P.ldx % 4 != 0 || P.Kin != 8 * NG)) return LRG_EINVAL - 31;
        if (i > 0 && (L.N != a.p[0].L[0].N || P.Kin != a.p[0].Kin || (L.flags & LRG_FL_RELU) != (a.p[0].L[0].flags & LRG_FL_RELU))) return LRG_EINVAL - 32;
        if (P.rows == 0 || (P.rows >> 5) >= (1L << 26)) return LRG_EINVAL - 34;
        b.x[i] = P.x; b.w[i] = L.w; b.bias[i] = L.bias; b.y[i] = L.gout; b.rows[i] = P.rows; b.ldx[i] = P.ldx; b.rows_per_inst[i] = P.rows_per_inst;
        b.inst_bias[i] = (L.flags & LRG_FL_INST_BIAS) != 0;
    }
    const int ncg = a.p[0].L[0].N / (32 * CT);
    if (ncg & (ncg - 1)) return LRG_EINVAL - 33;
    for (b.cg_shift = 0; (1 << b.cg_shift) < ncg; ++b.cg_shift) {}
    b.K = a.p[0].Kin; b.N = a.p[0].L[0].N; b.relu = (a.p[0].L[0].flags & LRG_FL_RELU) != 0; b.nprob = nprob;
    static const int dbg = getenv("LRG_STREAM_DBG") ? atoi(getenv("LRG_STREAM_DBG")) : 0;
    b.dbg = dbg;
    const size_t lds = (size_t)CT * NG * 1024;
    auto kern = lrg_stream_layer_lds_kernel<NG, CT, D, OCC, FIRST, W, RT>;
    static bool attr_done[LRG_MAX_DEVICES] = {};      // per instantiation, per device
    const int dev = lrg_current_device();
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
        attr_done[dev] = true;
    }
    // every CU full of workgroups (LDS: 160 KB; registers: OCC wavefronts per SIMD), a multiple of 8 XCDs x units
    const int units = nprob * ncg;
    int per_cu = (int)(160 * 1024 / (lds + 1024));
    if (per_cu > OCC * 4 / W) per_cu = OCC * 4 / W;
    if (per_cu < 1) return LRG_EINVAL - 35;
    long maxt = 0;
    for (int i = 0; i < nprob; ++i) maxt = (a.p[i].rows >> 5) / RT > maxt ? (a.p[i].rows >> 5) / RT : maxt;
    int m = 256 * per_cu / (8 * units);
    const int need = (int)((maxt + W * 8 - 1) / (W * 8));      // walks of one tile per wavefront cover everything with this many workgroups per unit and XCD
    if (m > need) m = need;
    if (m < 1) m = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * units * m)), dim3(64 * W), lds, st, b);
    LRG_LAUNCH_CHECK();
    return 0;
}

template <int NG, int CT, int D, int OCC, bool FIRST>
static int lrg_stream_layer_launch(const LrgFusedArgs &a, int nprob, hipStream_t st) {
    long items = 0;
    LrgStreamArgs b = {};
    for (int i = 0; i < nprob; ++i) {
        const LrgFusedProb &P = a.p[i];
        const LrgFusedLayer &L = P.L[0];
        if (P.rows % 32 != 0 || P.rows_per_inst % 32 != 0 || L.N % (32 * CT) != 0 || L.ng != NG || !L.gout || !L.w || !L.bias) return LRG_EINVAL - 30;
        if (!FIRST && (P.ldx % 4 != 0 || P.Kin != 8 * NG)) return LRG_EINVAL - 31;
        if (i > 0 && (L.N != a.p[0].L[0].N || P.Kin != a.p[0].Kin || (L.flags & LRG_FL_RELU) != (a.p[0].L[0].flags & LRG_FL_RELU))) return LRG_EINVAL - 32;
        items += (P.rows >> 5) * (L.N / (32 * CT));
        b.x[i] = P.x; b.w[i] = L.w; b.bias[i] = L.bias; b.y[i] = L.gout; b.rows[i] = P.rows; b.ldx[i] = P.ldx; b.rows_per_inst[i] = P.rows_per_inst;
        b.inst_bias[i] = (L.flags & LRG_FL_INST_BIAS) != 0;
    }
    if (items == 0) return 0;
    const int ncg = a.p[0].L[0].N / (32 * CT);
    if (ncg & (ncg - 1)) return LRG_EINVAL - 33;
    for (b.cg_shift = 0; (1 << b.cg_shift) < ncg; ++b.cg_shift) {}
    if (items >= (1L << 31)) return LRG_EINVAL - 34;
    b.K = a.p[0].Kin; b.N = a.p[0].L[0].N; b.relu = (a.p[0].L[0].flags & LRG_FL_RELU) != 0; b.nprob = nprob;
    hipLaunchKernelGGL((lrg_stream_layer_kernel<NG, CT, D, OCC, FIRST>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, b);
    LRG_LAUNCH_CHECK();
    return 0;
}
