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
    // packed rows (lrg_forward_packed): x holds the distinct rows of ALL instances back to back, *nrows of them; row r belongs
    // to instance row_inst[r].  Tiles are 32 consecutive packed rows and may span instances: the max-pool and the
    // per-instance bias are applied per run of equal row_inst inside the tile.
    const int *nrows;    // nullable: device count of packed rows (non-NULL selects the packed formulation)
    const int *row_inst; // [capacity] instance of each packed row
    const float *center; // nullable (packed): [instances,16] per-instance centre; the staged value of row r, column c is
                         // x[r,c] - center[row_inst[r]*16 + c]  (the rows are stored uncentred, test_region_grow.py:243-247 applied here)
    // medians riding in the same launch (LrgFusedArgs.nmed > 0): the centre of instance i, channel c arrives as the tagged word
    // ctag[i*16 + c] = (tags[2*i + 1] << 32) | float bits, written by this launch's median workgroups; channels outside cmask
    // are not centred
    const unsigned long long *ctag;
    const int32_t *tags;
    unsigned cmask;
    long rows;
    int ldx, Kin, rows_per_inst, pool_stride, nlayers, zero_count;
    LrgFusedLayer L[LRG_FUSED_MAXL];
};

// The nine channel medians of every slot's region (test_region_grow.py:241) computed by the FIRST workgroups (a few per slot,
// lrg_fused_median_workgroups) of the packed branch launch instead of a launch of their own: they never wait, the tile workgroups behind them (dispatched
// in order, so every median workgroup is already running or done) pick each centre up as one tagged 64-bit word.
struct LrgFusedMedians {
    const LrgSlot *slots;
    const LrgRoom *rooms;
    const int32_t *big;          // [n_slots,2]: (rows prepared this iteration, tag) written by lrg_front_greedy_kernel
    float *center;               // [n_slots,16] plain copy for the next launches (mask update, :271,:275)
    unsigned long long *ctag;    // [n_slots,16]
    int64_t *phase_ticks;        // nullable
    int n_slots, ncentred, F;
};

struct LrgFusedArgs {
    LrgFusedProb p[2];
    int nprob;           // set by the packed launchers: problems interleaved in a one-dimensional grid
    int nmed;            // median workgroups in front of the tiles (0: none)
    int few;             // packed launches: 1 = few tiles (at most ~one per CU): they are accounted 256 VGPRs, two tiles per CU at most
    LrgFusedMedians med;
};

int lrg_fused_branches(const LrgFusedArgs &a, int nprob, hipStream_t st);
int lrg_fused_heads(const LrgFusedArgs &a, int nprob, hipStream_t st);
// packed-row variants: P.rows = row capacity (multiple of 32), P.nrows / P.row_inst set
int lrg_fused_branches_packed(const LrgFusedArgs &a, int nprob, hipStream_t st);
int lrg_fused_heads_packed(const LrgFusedArgs &a, int nprob, hipStream_t st);

int lrg_fused_median_workgroups(int n_slots);       // median workgroups in front of the tiles of a packed branch launch

// lrg_forward_packed with the medians in the branch launch (lrg_net.hip; called by lrg_grow_step_packed)
int lrg_forward_packed_medians(const LrgWeights *w, const float *x_in, const float *x_nb, const int32_t *row_inst_in,
                               const int32_t *row_inst_nb, int32_t *nrows, int32_t *nrows_heads, int n_inst, int row_cap,
                               float *add_logits, float *rmv_logits, void *workspace, size_t workspace_bytes,
                               const LrgFusedMedians *med, hipStream_t st);
