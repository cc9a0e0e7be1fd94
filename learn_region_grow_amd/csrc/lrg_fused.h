// Internal interface between lrg_net.hip (lrg_forward) and lrg_fused.hip (fused stack kernels).
#pragma once
#include <hip/hip_runtime.h>
#include "lrg_common.h"

#define LRG_FUSED_MAXL 6
#define LRG_FL_RELU 1
#define LRG_FL_INST_BIAS 2   // bias is [instances, N]: the hoisted pooled-feature product of a head
#define LRG_FL_POOL 4        // column max over the instance's rows -> pool
#define LRG_FL_KEEP 8        // output stays in LDS as the next layer's input
#define LRG_FL_INPLACE 16     // output overlays this layer's own input buffer (last layer, one column block)

struct LrgFusedLayer {
    const float *w;      // MFMA-operand image of the [K,N] kernel (lrg_pack_weights layout): [N/32][ng][64 lanes][4]
    const float *bias;
    float *gout;         // nullable: copy of the output in HBM, [rows,N]
    int K, N, ng, flags; // ng = ceil(K/8) k-groups
};

struct LrgFusedProb {
    const float *x;      // [rows,Kin], row stride ldx
    float *pool;         // pool[(row / rows_per_inst) * pool_stride + col], zero-filled by the caller
    const float *fw;     // nullable: final [C,2] layer applied to the last LDS-resident activations
    const float *fb;
    float *fout;         // [rows,2]
    const int *valid;    // nullable, [instances]: only the first valid[i] rows of instance i are evaluated (0 = skip)
    const int *tile_count; // nullable pair: *tile_count live tiles, tile_list[b] = instance * 64 + tile of workgroup b
    const int *tile_list;
    float *zero_pool;    // nullable: after the stack, the tile-0 workgroup of instance i clears zero_pool[i*zero_count .. +zero_count)
    float *pool_rows;    // nullable (single-instance tiles of the free-running kernel): the tile's column maxima of the pooled layer go to
    int pool_rows_stride;  // pool_rows[inst * pool_rows_stride + tile * N + col] as ONE row of 16-byte write-through stores instead of N atomicMax
                         // on pool; whoever consumes the pooled feature takes the maximum over the instance's tiles (lrg_async.inl: the units)
    // packed rows (lrg_forward_packed): x holds the distinct rows of ALL instances back to back, *nrows of them; row r belongs
    // to instance row_inst[r].  Tiles are 32 consecutive packed rows and may span instances: the max-pool and the
    // per-instance bias are applied per run of equal row_inst inside the tile.
    const int *nrows;    // nullable: device count of packed rows (non-NULL selects the packed formulation)
    const int *row_inst; // [capacity] instance of each packed row
    const float *center; // nullable (packed): [instances,16] per-instance centre; the staged value of row r, column c is
                         // x[r,c] - center[row_inst[r]*16 + c]  (the rows are stored uncentred, test_region_grow.py:243-247 applied here)
    long rows;
    int ldx, Kin, rows_per_inst, pool_stride, nlayers, zero_count;
    LrgFusedLayer L[LRG_FUSED_MAXL];
};

struct LrgFusedArgs {
    LrgFusedProb p[2];
    int nprob;           // set by the packed launchers: problems interleaved in a one-dimensional grid
    int few;             // packed launches: 1 = few tiles (at most ~one per CU): they are accounted 256 VGPRs, two tiles per CU at most
};

int lrg_fused_branches(const LrgFusedArgs &a, int nprob, hipStream_t st);
int lrg_fused_heads(const LrgFusedArgs &a, int nprob, hipStream_t st);
// one layer per launch (lrg_forward, LRG_FWD_STREAM_TILES): 1-layer problems with gout set and no KEEP
int lrg_fused_layer(const LrgFusedArgs &a, int nprob, hipStream_t st);
// packed-row variants: P.rows = row capacity (multiple of 32), P.nrows / P.row_inst set
int lrg_fused_branches_packed(const LrgFusedArgs &a, int nprob, hipStream_t st);
int lrg_fused_heads_packed(const LrgFusedArgs &a, int nprob, hipStream_t st);

// hoisted pooled-feature product of the heads' first layer: hb[b,c] = bias[c] + sum_k pooled[b,k] w[k,c]  (lrg_net.hip)
struct LrgGemvArgs {
    const float *pooled;
    const float *w[2];
    const float *bias[2];
    float *hb[2];
    int ldw, B, P, C;
    int *cnt_src, *cnt_dst;   // nullable pair (packed rows): block (0,0,0) copies cnt_src[0..1] to cnt_dst[0..1] and zeroes cnt_src --
                              // the branch kernels are done with the row counts, the heads read the copy, the next front kernel
                              // allocates from zero again
};

// The descriptors of a packed evaluation (two branch stacks, pooled product, two head stacks) without launching anything: for the
// free-running region-grow kernel, whose tile teams run them task by task (lrg_async.inl).  Needs w->packed.
int lrg_packed_problems(const LrgWeights *w, const float *x_in, const float *x_nb, const float *center, const int32_t *row_inst_in,
                        const int32_t *row_inst_nb, int32_t *nrows, int n_inst, int row_cap, float *add_logits, float *rmv_logits,
                        void *workspace, size_t workspace_bytes, LrgFusedArgs *branches, LrgGemvArgs *gemv, LrgFusedArgs *heads);
// float offsets in the packed workspace of the two [row_cap, conv_ch[n_conv - 2]] row arrays a wave-branch launch hands from its PREFIX to its POOL tasks
int lrg_packed_conv3_view(const LrgWeights *w, int n_inst, int row_cap, size_t offset_floats[2]);
