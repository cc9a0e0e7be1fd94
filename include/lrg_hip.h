/* lrg_hip.h -- C-ABI of liblrg_hip.so: the MI355X (gfx950) LRGNet region-grow hot path.
 *
 * Conventions (all entry points):
 *   - pointers are DEVICE pointers unless a parameter is documented as host; buffers are caller-owned;
 *   - tensors are dense row-major; kernels of the network are [Cin,Cout] row-major, i.e. the TF
 *     variable of shape [1,Cin,Cout] (learn_region_grow_util.py:107) with the leading 1 dropped;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream) and the
 *     call returns without synchronising; no entry point allocates or frees device memory;
 *   - return value: 0 on success, -(hipError_t) on a HIP failure, LRG_EINVAL (-1000 - n) on a bad
 *     argument; re-entrant, no global state.
 *
 * Every entry point names the reference interface it replaces (file:line under the reference repo).
 */
#ifndef LRG_HIP_H
#define LRG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRG_ABI_VERSION 10
#define LRG_EINVAL (-1000)
#define LRG_ERESIDENCY (-1100)  /* lrg_grow_async: the launch's workgroups cannot all be resident at once on this stream / device (see there) */

#define LRG_MAX_CONV 5 /* branch layers: lite=0 -> 5, lite=1 -> 2, lite=2 -> 3  (learn_region_grow_util.py:77-85) */
#define LRG_MAX_HEAD 3 /* head layers incl. the final 2-wide one: lite=0 -> 3, lite=1 -> 2, lite=2 -> 3        */

int lrg_abi_version(void);
/* Name of the code object's target ("gfx950"); a build sanity hook for the loader. */
const char *lrg_target_arch(void);
/* sizeof() of the ABI structs as compiled (0 LrgWeights, 1 LrgRoom, 2 LrgSlot, 3 LrgGrowParams, 4 LrgStepBuffers, 5 LrgPackedBuffers, 6 LrgBeamGroup, 7 LrgAsyncBuffers, 8 LrgFillJob), so that a
 * foreign-language binding can verify its mirror of the layout at load time. */
size_t lrg_struct_size(int which);

/* ------------------------------------------------------------------------------------------------
 * LrgNet forward   (replaces: sess.run([net.add_output, net.remove_output], ...) at
 * test_region_grow.py:257-258 on the graph built by learn_region_grow_util.py:75-162)
 * ---------------------------------------------------------------------------------------------- */
typedef struct LrgWeights {
    int32_t feature_size;            /* F: 13 (6/9/12 variants, test_region_grow.py:72-77)                 */
    int32_t n_conv;                  /* branch layers                                                       */
    int32_t n_head;                  /* head layers including the final [C,2] layer                         */
    int32_t reserved;
    int32_t conv_ch[LRG_MAX_CONV];   /* CONV_CHANNELS                                                       */
    int32_t head_ch[LRG_MAX_HEAD];   /* CONV2_CHANNELS followed by 2                                        */
    const float *inlier_w[LRG_MAX_CONV];   /* lrg_kernel{i}          [Cin,Cout]   :107                      */
    const float *inlier_b[LRG_MAX_CONV];   /* lrg_bias{i}            [Cout]       :108                      */
    const float *neighbor_w[LRG_MAX_CONV]; /* lrg_neighbor_kernel{i}              :115                      */
    const float *neighbor_b[LRG_MAX_CONV]; /* lrg_neighbor_bias{i}                :116                      */
    const float *add_w[LRG_MAX_HEAD];      /* lrg_add_kernel{j}; j=0 is [2*C_last + conv_ch[1], C]  :139,:145 */
    const float *add_b[LRG_MAX_HEAD];      /* lrg_add_bias{j}                                       :140,:146 */
    const float *rmv_w[LRG_MAX_HEAD];      /* lrg_remove_kernel{j}                                  :152,:158 */
    const float *rmv_b[LRG_MAX_HEAD];      /* lrg_remove_bias{j}                                    :153,:159 */
    const void *packed;                    /* nullable: lrg_pack_weights() image of the kernels above (MFMA operand order).
                                              NULL: lrg_forward (LRG_FWD_FUSED) re-packs into its workspace on every call,
                                              which is always correct; non-NULL: the caller vouches it is current.   */
} LrgWeights;

/* The fused kernels read the [Cin,Cout] kernels in the order their MFMA B operands want them (one 16-byte load per lane
 * per 8 rows of Cin instead of four strided 4-byte loads).  lrg_pack_weights writes that image (device memory,
 * lrg_packed_weights_bytes(w) bytes, 256-byte aligned) from the TF-layout pointers in `w`; store its address in
 * w->packed after a restore (test_region_grow.py:92-93) and repeat after any change of the variables. */
size_t lrg_packed_weights_bytes(const LrgWeights *w);
int lrg_pack_weights(const LrgWeights *w, void *packed, size_t packed_bytes, void *stream);

#define LRG_FWD_FUSE_POOL 1u /* layer-streamed path: fold the max-pool (:122-123) into the last branch layer's epilogue */
#define LRG_FWD_FUSED 2u     /* whole branch / whole head per 64-row tile in one kernel each: activations stay in LDS,
                                only conv[1], the pooled maxima and the logits reach HBM (3 launches per call)         */
#define LRG_FWD_KEEP_ACTS 4u /* with LRG_FWD_FUSED: also copy every intermediate into the workspace (parity tests)     */
#define LRG_FWD_TILE_LISTS 16u /* lrg_forward_rows + LRG_FWD_FUSED: the workspace's tile-list block (view kind 6) was filled by
                                  lrg_prepare for exactly these row counts: only the listed 32-row tiles are launched as
                                  working workgroups, back to back, so they spread evenly over the compute units (skipping
                                  by row count alone leaves the survivors wherever the dead tiles happened to sit).      */
#define LRG_ROW_TILE 32        /* rows per tile of those lists */
#define LRG_FWD_STREAM_TILES 32u /* layer-streamed path (no LRG_FWD_FUSED): every layer launch runs on the fused stacks' tile (a 1-layer stack: input rows staged once
                                 for all column blocks, weights in operand order, output from the accumulators to HBM) instead of lrg_pointwise_mfma_kernel's
                                 64 x 64 tiles; same layer-by-layer formulation and HBM traffic model (ABI 9)                                                     */
#define LRG_FWD_POOL_ZEROED 8u /* with LRG_FWD_FUSED: the workspace was zero-filled once by the caller and is only ever used
                                by calls carrying this flag -- the pooled-feature block is then zero on entry and is
                                left zero on return (cleared by the head kernel), which saves the per-call memset.
                                Precondition on rows: rows_nb[b] > 0 whenever rows_in[b] > 0.                            */

/* Bytes of scratch lrg_forward needs for a batch of B instances (host-side arithmetic only). */
size_t lrg_forward_workspace_bytes(const LrgWeights *w, int B, int n_inlier, int n_neighbor);

/* inlier [B,n_inlier,F], neighbor [B,n_neighbor,F] -> add_logits [B,n_neighbor,2] (net.add_output :149),
 * rmv_logits [B,n_inlier,2] (net.remove_output :162).  `w` is a HOST struct holding device pointers. */
int lrg_forward(const LrgWeights *w, const float *inlier, const float *neighbor, int B, int n_inlier,
                int n_neighbor, float *add_logits, float *rmv_logits, void *workspace, size_t workspace_bytes,
                unsigned flags, void *stream);

/* lrg_forward restricted to the leading rows of each instance.  rows_in / rows_nb ([B] device int32, both or
 * neither NULL): only rows [0, rows_in[b]) of inlier[b] and [0, rows_nb[b]) of neighbor[b] are evaluated.  The caller
 * guarantees that every later row is a copy of one of those (the reference pads small sets by duplication,
 * test_region_grow.py:240,:252), so their logits equal the source row's bit for bit and the max-pool (:122-123) is
 * unchanged; logits of rows that are not evaluated are left unwritten.  A count of 0 skips the instance.
 * Needs LRG_FWD_FUSED (whole 32-row tiles of copies are skipped). */
int lrg_forward_rows(const LrgWeights *w, const float *inlier, const float *neighbor, int B, int n_inlier,
                     int n_neighbor, const int32_t *rows_in, const int32_t *rows_nb, float *add_logits,
                     float *rmv_logits, void *workspace, size_t workspace_bytes, unsigned flags, void *stream);

/* Workspace introspection for layer-by-layer parity tests: float offset / element count of a named
 * intermediate inside `workspace`.  kind: 0 conv[i] (inlier), 1 neighbor_conv[i], 2 pooled [B,2*C_last],
 * 3 add head hidden[i], 4 remove head hidden[i], 5 scratch (64 floats, reserved), 6 tile lists (int32: [0] inlier tile
 * count, [1] neighbour tile count, then B*ceil(n_inlier/32) inlier entries and B*ceil(n_neighbor/32) neighbour entries,
 * entry = instance * 64 + tile).
 * Returns 0, or LRG_EINVAL. */
int lrg_forward_workspace_view(const LrgWeights *w, int B, int n_inlier, int n_neighbor, int kind, int index,
                               size_t *offset_floats, size_t *count_floats);

/* One per-point layer  y = act(x @ w + bias)   (tf.nn.conv1d k=1 + bias_add + relu, :109-111).
 * x [rows,cin] (row stride ldx), w [cin,cout] (row stride ldw), y [rows,cout].
 * bias is [cout], or -- when rows_per_instance > 0 and bias_instance_stride > 0 -- one bias row per
 * instance: bias[(row / rows_per_instance) * bias_instance_stride + col] (the hoisted pooled-feature
 * product of the heads, :128-141).  pool_out (nullable, needs relu) receives
 * max over each instance's rows: pool_out[(row / rows_per_instance) * pool_stride + col]; it must be
 * zero-filled by the caller beforehand. */
int lrg_pointwise_layer(const float *x, int ldx, const float *w, int ldw, const float *bias, float *y, long rows,
                        int cin, int cout, int relu, int rows_per_instance, int bias_instance_stride,
                        float *pool_out, int pool_stride, void *stream);

/* Column max over each instance's rows: out[b*out_stride + c] = max_r x[b,r,c]   (tf.reduce_max axis=1, :122-123) */
int lrg_segmax(const float *x, float *out, int B, int rows, int C, int out_stride, void *stream);

/* Hoisted pooled-feature product of a head's first layer:
 * hb[b, c] = bias[c] + sum_k pooled[b,k] * w[k,c],  k < P   (the tiled part of the concat at :128-135) */
int lrg_head_pool_gemv(const float *pooled, const float *w, int ldw, const float *bias, float *hb, int B, int P,
                       int C, void *stream);

/* Final head layer, no ReLU: logits[r, 0..1] = h[r,:] @ w[C,2] + bias[2]   (:145-149, :158-162) */
int lrg_head_final(const float *h, const float *w, const float *bias, float *logits, long rows, int C, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Region-grow loop state (replaces the module-level state of test_region_grow.py:175-205)
 * ---------------------------------------------------------------------------------------------- */
enum {
    LRG_IDLE = 0,            /* slot not bound to work                                                     */
    LRG_ACTIVE = 1,          /* region growing                                                             */
    LRG_STOP_NONEIGHBOR = 2, /* test_region_grow.py:233-235                                                */
    LRG_STOP_NOEXPAND = 3,   /* :304-306                                                                   */
    LRG_STOP_STUCK = 4,      /* :294-297                                                                   */
    LRG_STOP_EMPTY = 5,      /* mask became empty; the reference raises at :292 -- defined as a stop here  */
    LRG_STOP_MAXSTEPS = 6,   /* optional safety cap (off when max_region_steps <= 0)                        */
    LRG_DONE = 7,            /* the slot's room has no unvisited seed left                                  */
    LRG_WAIT = 8,            /* finished its restarts, waiting for its group; also the state a host binds a
                                fresh group in (with seed = -1) so that lrg_advance picks the first seed    */
    LRG_PENDING = 9          /* lrg_grow_async with LrgAsyncBuffers.speculate > 1: the slot's region has stopped growing
                                (last_reason says why) and waits for the regions of the room's earlier seeds to be
                                committed first (test_region_grow.py:186-188 visits the seeds one after the other)    */
};

#define LRG_LOG_WORDS 8
/* One room, device-resident.  Arrays are per equalised point (n of them). */
typedef struct LrgRoom {
    const float *points;     /* [n,F] features as stacked at test_region_grow.py:165-172                    */
    const int32_t *voxels;   /* [n,3] rint(xyz/resolution)  (:175)                                          */
    const int32_t *obj_id;   /* [n] ground-truth instance id (:116), only for the GT flags (:230-231)        */
    const int32_t *order;    /* [n] seed order = argsort(curvatures) (:183)                                  */
    uint8_t *visited;        /* [n] (:178)                                                                  */
    int32_t *label;          /* [n] cluster_label before fill-in (:176)                                      */
    const uint64_t *hash_keys; /* voxel hash table (open addressing), hash_mask+1 entries                    */
    const int32_t *hash_vals;
    int32_t *region_log;     /* [n,LRG_LOG_WORDS] per committed seed: seed, steps, points, reason, labeled, restart, and the
                                number of sample slots whose argmax matched input_add / input_remove at the region's last
                                evaluated step (add_acc, remove_acc of learn_region_grow_util.py:175,180 times 512; -1: none) */
    int32_t n;
    int32_t hash_mask;
    int32_t next_cluster_id; /* (:177)                                                                      */
    int32_t seed_cursor;     /* position in `order` of the next candidate seed (:186-188)                    */
    int32_t n_regions;
    int32_t done;
    int32_t room_id;         /* RNG stream key and log tag                                                   */
    int32_t pad;
    const uint32_t *pvox;    /* nullable: [n] voxels relative to vox_origin packed x | y << 11 | z << 22 (lrg_voxel_pack); the
                                arrays visited / pvox must then start 16-byte aligned (word-wide loads of 4 points)   */
    int32_t vox_origin[3];   /* per-axis minimum voxel of the room                                           */
    int32_t chan_stride;     /* floats between two channels of chan_major                                    */
    const float *chan_major; /* nullable: the centred channels 0, 1, 6 .. F-1 (:243-247) channel-major,
                                chan_major[y * chan_stride + i] == points[i * F + (y < 2 ? y : y + 4)]: a region's points are
                                mostly runs of consecutive indices, so the median's keys (:241) come from a few dense lines
                                instead of one 4-byte word out of every F-float row                          */
    const int32_t *vgrid;    /* nullable (needs pvox / vox_origin): dense voxel grid of the room, x fastest,
                                vgrid[((vz - oz) * vgrid_dim[1] + (vy - oy)) * vgrid_dim[0] + (vx - ox)] = index of the point in
                                that voxel or -1 (lrg_voxel_grid_build).  The box query of a region (:221-229) then reads the cells
                                of its dilated box -- rows of consecutive ints -- instead of every point of the room, and a voxel
                                lookup (:273-287) is one load instead of a probe chain: 4 bytes per voxel of the room's bounding
                                box, a few MB per room out of 288 GB                                          */
    int32_t vgrid_dim[3];    /* voxels per axis of the room's bounding box                                   */
    int32_t pad2;
} LrgRoom;

/* One growing region instance.  A *group* of `group_size` consecutive slots shares one room and one seed
 * (random restarts, test_random_restart.py:169-197); greedy growing is group_size = 1, restarts = 1. */
typedef struct LrgSlot {
    uint8_t *cur;            /* [cap] currentMask (:197)                                                    */
    uint8_t *best;           /* [cap] best restart mask so far (restart :175-177); unused when restarts==1   */
    int32_t *cur_idx;        /* [cap] compacted indices of current points, index order (:221)                */
    int32_t *cand_idx;       /* [cap] compacted indices of expand candidates (:229)                          */
    int32_t room;            /* index into rooms[], -1 = none                                                */
    int32_t status;
    int32_t seed;            /* seed point (:186)                                                            */
    int32_t restart;         /* restart ordinal of the grow in progress                                      */
    int32_t step;            /* steps of the grow in progress (RNG counter)                                  */
    int32_t steps_total;     /* `steps` of the reference: never reset between restarts of a seed             */
    int32_t stuck;           /* (:204)                                                                       */
    int32_t nc;              /* len(currentPoints)                                                           */
    int32_t ne;              /* len(expandPoints)                                                            */
    int32_t updated;         /* (:278)                                                                       */
    int32_t count;           /* sum(currentMask) after the last update                                       */
    int32_t best_count;      /* restart_score of `best` (--scoring np), -1 = none                            */
    int32_t best_restart;
    int32_t last_reason;
    int32_t mn[3], mx[3];    /* minDims / maxDims (:199-200)                                                 */
    int32_t seq_mn[3], seq_mx[3]; /* seqMinDims / seqMaxDims (:201-202)                                      */
    int32_t target;          /* obj_id[seed] (:190)                                                          */
    int32_t pad;             /* 1 while cur_idx / cand_idx describe the current mask and bbox (set by lrg_box_query)  */
    int32_t *chunk_cnt;      /* [2 * ceil(cap / LRG_SCAN_CHUNK)] per-chunk (current, candidate) counts of lrg_box_query */
    int32_t scan_cnt;        /* lrg_bbox_stop scratch: members / min / max voxel of the updated mask, consumed (and    */
    int32_t scan_mn[3];      /*   reset) by lrg_advance                                                               */
    int32_t scan_mx[3];
    int32_t query;           /* lrg_box_query scratch: 1 if this call re-derives the slot's lists                     */
    int32_t acc_add;         /* sample slots whose argmax(add logits) == input_add at the last evaluated step (:175), -1 none */
    int32_t acc_rmv;         /* likewise for the remove head (:180)                                                     */
    double ml_score;         /* --scoring ml: log-likelihood of the masks sampled in the restart in progress (restart :251-271) */
    double ml_best;          /* ... of the banked restart                                                                */
    int32_t spec_pos;        /* speculation (ABI 9): position of the slot's seed in the room's seed order, INT32_MAX = no region in progress */
    int32_t spec_flags;      /* bit 0: the region in progress is void -- a region committed before it took a point inside a box it
                                had queried (:222-228 would have seen that point as visited); it is dropped and grown again         */
} LrgSlot;

#define LRG_SCAN_CHUNK 4096  /* points per workgroup of the chunked mask scans */

typedef struct LrgGrowParams {
    float resolution;        /* 0.1 (test_region_grow.py:27)                                                 */
    int32_t feature_size;
    int32_t n_inlier;        /* NUM_INLIER_POINT   (:22)                                                     */
    int32_t n_neighbor;      /* NUM_NEIGHBOR_POINT (:23)                                                     */
    int32_t cluster_threshold; /* 10 (:30)                                                                   */
    int32_t restarts;        /* NUM_RESTARTS; 1 = greedy                                                     */
    int32_t group_size;      /* slots per group; restarts are dealt round-robin over the group's slots       */
    int32_t max_region_steps;/* <= 0: no cap                                                                 */
    uint32_t rng_seed;       /* counter-stream key word 0 (key word 1 = room_id)                             */
    int32_t policy;          /* 0 net (:266-267), 1 threshold (:264-265), 2 ground truth (:268-269)          */
    int32_t scoring;         /* restart score (test_random_restart.py:171-174): 0 = 'np' points of the mask, 1 = 'ml' summed
                                log-likelihood of the sampled add / remove masks, one scalar per restart (upstream resets the
                                accumulator to a list at :194-196, which breaks it from the second restart on; implemented as
                                evidently intended).  'ml' is computed by lrg_grow_step_packed only.                        */
} LrgGrowParams;

/* Bind the slots [first_slot, first_slot + group_size) to room `room` (-1: leave them idle), on the device: every slot
 * waits with seed -1, so the next advance picks the room's first seed (:186-188).  reset_room != 0 first returns the room to
 * its pristine state (visited / labels cleared, cursor at 0, :176-178); clear_masks != 0 clears the slots' masks over the
 * room they are leaving.  One launch, no host-to-device copy: safe to issue while other streams keep the host busy. */
int lrg_bind_group(LrgSlot *slots, LrgRoom *rooms, int first_slot, int group_size, int room, int reset_room, int clear_masks,
                   void *stream);

/* voxels[i,0..2] = rint(points[i,0..2] / resolution)   (test_region_grow.py:175) */
int lrg_voxelize(const float *points, int n, int F, float resolution, int32_t *voxels, void *stream);

/* pvox[i] = (vx - ox) | (vy - oy) << 11 | (vz - oz) << 22 for the room's voxels: one word per point for the box query and the
 * bounding-box passes of lrg_grow_step_packed.  (ox, oy, oz) = the per-axis minimum voxel; the room must span at most
 * 2048 x 2048 x 1024 voxels (the caller checks; out-of-range coordinates are clamped and set *overflow_flag, device int). */
int lrg_voxel_pack(const int32_t *voxels, int n, int ox, int oy, int oz, uint32_t *pvox, int32_t *overflow_flag, void *stream);

/* Dense voxel grid of a room (LrgRoom.vgrid): grid[gx * gy * gz] (x fastest) is filled with -1 and receives the index of the
 * point of each occupied voxel; (ox, oy, oz) = the per-axis minimum voxel, (gx, gy, gz) = the extent of the bounding box. */
int lrg_voxel_grid_build(const int32_t *voxels, int n, int ox, int oy, int oz, int gx, int gy, int gz, int32_t *grid, void *stream);

/* Build the room's voxel -> point-index table (replaces the tuple sets of :273,:277,:283-286).
 * keys must hold hash_mask+1 entries; *dup_flag (device int, zeroed by the caller) is set when two points
 * share a voxel (the path assumes an equalised room, :125-134). */
int lrg_voxel_hash_build(const int32_t *voxels, int n, uint64_t *keys, int32_t *vals, int hash_mask,
                         int32_t *dup_flag, void *stream);

/* Scan of the updated mask (:292-293): member count and min / max voxel of every ACTIVE slot that has taken a mask
 * update (slot.updated >= 0), reduced into slot.scan_*.  The stop / stuck decision of :291-306 is taken from those
 * values at the start of the following lrg_advance.  max_points = capacity of the per-slot masks (grid sizing). */
int lrg_bbox_stop(LrgSlot *slots, const LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                  void *stream);

/* Commit finished seeds (:210-217 / restart :169-197), pick the next unvisited seed (:186-188) and reset
 * the group's slots (:197-204).  stats (device, LRG_STATS_WORDS x int64, nullable): see LrgStepBuffers. */
int lrg_advance(LrgSlot *slots, LrgRoom *rooms, int n_slots, const LrgGrowParams *params, int64_t *stats,
                void *stream);

/* Dilated voxel-box neighbour query + ordered compaction (:221-235): fills cur_idx/nc, cand_idx/ne; a slot
 * with ne == 0 stops with LRG_STOP_NONEIGHBOR. */
int lrg_box_query(LrgSlot *slots, const LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                  void *stream);

/* center[s, c] = median over the slot's current points of channel c for c in {0,1} U [6,F), 0 elsewhere
 * (numpy.median, :241; only those channels are used, :243-247).  center is [n_slots,16]. */
int lrg_median(const LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params, float *center,
               void *stream);

/* Counter-stream subset sampling (:237-240, :249-252): positions into cur_idx / cand_idx,
 * sample_in [n_slots,n_inlier], sample_nb [n_slots,n_neighbor]. */
int lrg_sample(const LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params,
               int32_t *sample_in, int32_t *sample_nb, void *stream);

/* Gather + centre (:242-254): inlier [n_slots,n_inlier,F], neighbor [n_slots,n_neighbor,F];
 * gt_remove / gt_add (nullable) receive input_remove / input_add (:248,:254).
 * rows_in / rows_nb (nullable, [n_slots]) receive the number of distinct leading rows of each stacked set:
 * min(nc, n_inlier) / min(ne, n_neighbor) for an active slot (the rest are duplicates, :240,:252), 0 otherwise. */
int lrg_gather_center(const LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params,
                      const int32_t *sample_in, const int32_t *sample_nb, const float *center, float *inlier,
                      float *neighbor, int32_t *gt_remove, int32_t *gt_add, int32_t *rows_in, int32_t *rows_nb,
                      void *stream);

/* lrg_median + lrg_sample + lrg_gather_center in one pass per slot (counter stream): the preparation of a step,
 * test_region_grow.py:237-254.  Same outputs as the three calls. */
int lrg_prepare(const LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params, float *center,
                int32_t *sample_in, int32_t *sample_nb, float *inlier, float *neighbor, int32_t *gt_remove, int32_t *gt_add,
                int32_t *rows_in, int32_t *rows_nb, int32_t *tile_total, void *stream);
/* tile_total (nullable, device int32 block = workspace view kind 6 of an n_slots-instance forward): receives the lists of
 * live 32-row tiles for lrg_forward_rows(LRG_FWD_TILE_LISTS). */

/* Confidence + Bernoulli masks + voxel-set mask update (:262-288).
 * add_mask / rmv_mask (nullable, uint8 [n_slots,n]) : host-decided masks (reference-order RNG);
 * when NULL the masks are drawn on the device from the counter stream against
 * softmax(logits)[:,1].  Sets slot.updated, increments slot.step / steps_total and, when stats is
 * non-NULL, stats[2] (instance-steps taken).
 * sample_in / sample_nb (nullable): when given, logits were produced by lrg_forward_rows, i.e. only for the distinct
 * leading rows; slot j of a set with fewer points than slots then reads the logits of row sample[j] (its source). */
int lrg_mask_update(LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params,
                    const float *inlier, const float *neighbor, const float *center, const float *add_logits,
                    const float *rmv_logits, const int32_t *gt_remove, const int32_t *gt_add,
                    const uint8_t *add_mask, const uint8_t *rmv_mask, const int32_t *sample_in,
                    const int32_t *sample_nb, int64_t *stats, void *stream);

/* Device buffers of one lock-step iteration over n_slots slots (all caller-owned). */
typedef struct LrgStepBuffers {
    float *center;        /* [n_slots,16]                 */
    int32_t *sample_in;   /* [n_slots,n_inlier]           */
    int32_t *sample_nb;   /* [n_slots,n_neighbor]         */
    float *inlier;        /* [n_slots,n_inlier,F]         */
    float *neighbor;      /* [n_slots,n_neighbor,F]       */
    int32_t *gt_remove;   /* [n_slots,n_inlier]           */
    int32_t *gt_add;      /* [n_slots,n_neighbor]         */
    float *add_logits;    /* [n_slots,n_neighbor,2]       */
    float *rmv_logits;    /* [n_slots,n_inlier,2]         */
    void *workspace;      /* lrg_forward scratch          */
    size_t workspace_bytes;
    int64_t *stats;       /* LRG_STATS_WORDS x int64      */
    int32_t *rows_in;     /* [n_slots] rows to evaluate per slot (nullable pair: NULL = evaluate all 512+512 rows) */
    int32_t *rows_nb;
} LrgStepBuffers;

/* stats layout: [0] committed seeds, [1] rooms finished, [2] instance-steps taken, [3] hand-overs given up by lrg_grow_async (0 unless something is broken),
 * [4 + k % LRG_DONE_RING] = first slot of the k-th finished group (ring written by lrg_advance; the greedy front kernels add the
 * room index: slot | room << 32). */
#define LRG_DONE_RING 1020
#define LRG_STATS_WORDS (4 + LRG_DONE_RING)

/* One whole lock-step iteration with device-side (counter-stream) randomness -- the body of the
 * `while True` loop at test_region_grow.py:208-306 for every slot at once:
 *   lrg_bbox_stop; advance_rounds x (lrg_advance; lrg_box_query); lrg_prepare (= lrg_median + lrg_sample +
 *   lrg_gather_center); lrg_forward_rows; lrg_mask_update.
 * `weights` and `buffers` are HOST structs of device pointers. */
int lrg_grow_step(LrgSlot *slots, LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                  const LrgWeights *weights, const LrgStepBuffers *buffers, int advance_rounds, unsigned forward_flags,
                  void *stream);

/* ------------------------------------------------------------------------------------------------
 * Packed-row formulation of the same iteration (the default of the batched loop).
 *
 * A stacked set with fewer points than sample slots is padded by re-drawing its own rows (test_region_grow.py:240,:252);
 * LrgNet is row-wise up to the max-pool, which ignores duplicates.  Here only the DISTINCT rows of every slot are
 * gathered, back to back for all slots (one packed array per branch), and the network runs on dense 32-row tiles of
 * those arrays: a tile may hold rows of several slots; the max-pool (:122-123) and the hoisted per-instance bias
 * (:128-141) are applied per run of rows of one slot.  Logits come back per packed row; sample slot j of a padded set
 * reads the logits of its source row.  Results are bit-identical to lrg_grow_step.
 * ---------------------------------------------------------------------------------------------- */

/* LrgNet on packed rows.  x_in / x_nb [row_cap,F] hold nrows[0] / nrows[1] valid rows (device int32 pair);
 * center (nullable, [n_inst,16]): when given, the rows are stored UNCENTRED and row r enters the network as
 * x[r,c] - center[row_inst[r]*16 + c] (test_region_grow.py:243-247 applied while the rows are staged; center is 0 on the
 * channels the reference leaves alone);
 * row_inst_in / row_inst_nb [row_cap] name the instance (0 <= . < n_inst) of each row, rows of one instance being
 * contiguous; row_cap is a multiple of LRG_ROW_TILE.  add_logits [row_cap,2] (per neighbour row, net.add_output :149),
 * rmv_logits [row_cap,2] (per inlier row, net.remove_output :162).
 * nrows_heads (nullable, device int32 pair): when given, the row counts are handed over to it and nrows is zeroed
 * between the branch and the head kernels, so that the caller's next allocation pass starts from zero.
 * flags: LRG_FWD_POOL_ZEROED = the pooled block of the workspace (lrg_forward_packed_pooled_view) is zero on entry.
 * Needs the layer widths the fused kernels are built for (lite 0/1/2). */
size_t lrg_forward_packed_workspace_bytes(const LrgWeights *w, int n_inst, int row_cap);
int lrg_forward_packed_pooled_view(const LrgWeights *w, int n_inst, int row_cap, size_t *offset_floats, size_t *count_floats);
int lrg_forward_packed(const LrgWeights *w, const float *x_in, const float *x_nb, const float *center, const int32_t *row_inst_in,
                       const int32_t *row_inst_nb, int32_t *nrows, int32_t *nrows_heads, int n_inst, int row_cap,
                       float *add_logits, float *rmv_logits, void *workspace, size_t workspace_bytes, unsigned flags,
                       void *stream);

/* Device buffers of one packed iteration over n_slots slots (all caller-owned; counters zero before the first call). */
typedef struct LrgPackedBuffers {
    float *center;          /* [n_slots,16]                                                                  */
    int32_t *sample_in;     /* [n_slots,n_inlier]   sample positions (:237-240)                              */
    int32_t *sample_nb;     /* [n_slots,n_neighbor] (:249-252)                                               */
    float *x_in;            /* [row_cap,F] packed distinct inlier rows of all slots (:245-247)               */
    float *x_nb;            /* [row_cap,F] packed distinct neighbour rows (:243-244,:253)                    */
    int32_t *row_slot_in;   /* [row_cap] slot of each packed row                                             */
    int32_t *row_slot_nb;
    float *upd_in;          /* [n_slots,n_inlier,4]   per slot, for its distinct inlier rows j: columns 0..2 of the packed row and
                                input_remove of its point (:248) as 0.0 / 1.0 -- what the NEXT mask update (:262-288) needs of
                                the row, in storage of the slot's own (16-byte aligned).  The packed arrays are allocated from row 0
                                again by every launch: a slot whose workgroup starts late must not look for last iteration's rows there */
    float *upd_nb;          /* [n_slots,n_neighbor,4] the same for the neighbour rows and input_add (:254)    */
    float *rmv_logits;      /* [row_cap,2] per packed inlier row                                             */
    float *add_logits;      /* [row_cap,2] per packed neighbour row                                          */
    int32_t *slot_rows;     /* [n_slots,4] rows_in, rows_nb, first packed inlier row, first packed neighbour row */
    int32_t *counters;      /* [4] packed rows allocated (in, nb) and their hand-over copy for the heads      */
    void *workspace;        /* lrg_forward_packed_workspace_bytes(w, n_slots, row_cap), 256-byte aligned       */
    size_t workspace_bytes;
    int64_t *stats;         /* LRG_STATS_WORDS x int64, as LrgStepBuffers                                      */
    int32_t row_cap;        /* multiple of LRG_ROW_TILE, >= n_slots * max(n_inlier, n_neighbor) rounded up to 16: a slot's
                                rows are allocated in multiples of 8, the padding being copies of its last row    */
    int32_t rooms_have_pvox; /* 1: every room carries pvox (and 16-byte aligned visited / pvox): enables the single-launch greedy
                                front kernel with word-wide room scans                                          */
    int32_t *slot_big;      /* nullable with rooms_have_pvox = 0: [n_slots,2] zero-filled ONCE and owned by these buffers for good:
                                (rows prepared this iteration, running tag of the slot's centres) -- the medians come from a
                                second, (slot, channel)-parallel launch                                            */
    int64_t *phase_ticks;   /* nullable: [n_slots,2] accumulators of wall_clock64() ticks the slot's workgroup spent in (0) mask
                                update / stop decision / commit -- the reference's 'inlier' bucket, test_region_grow.py:260-306 --
                                and (1) box query / median / sampling / gather -- its 'neighbor' bucket, :219-254            */
} LrgPackedBuffers;

/* One lock-step iteration, packed rows: lrg_front_kernel (mask update of the previous evaluation :262-288, stop decision
 * :291-306, commit / next seed :186-217, box query :221-235, medians :241, sampling :237-252, gather :242-254) and
 * lrg_forward_packed -- five launches (front, medians, branch stacks, pooled GEMM, head stacks; six with restart groups: update,
 * group commit, query + medians + gather, then the three of the network).
 * Slot masks (LrgSlot.cur) must be 4-byte aligned;
 * rooms of up to 131072 points, n_inlier / n_neighbor <= 1024; larger: lrg_grow_step. */
int lrg_grow_step_packed(LrgSlot *slots, LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                         const LrgWeights *weights, const LrgPackedBuffers *buffers, void *stream);
/* The two halves of lrg_grow_step_packed as separate calls (so that a caller can put events between them):
 * lrg_front_step = everything up to the packed rows; then lrg_forward_packed(weights, x_in, x_nb, centre, row_slot_in,
 * row_slot_nb, counters, counters + 2, n_slots, row_cap, add_logits, rmv_logits, workspace, workspace_bytes,
 * LRG_FWD_POOL_ZEROED, stream) with centre = lrg_packed_rows_center(params, buffers): the buffers' centre array when the front
 * kernels store uncentred rows (greedy growing on rooms with packed voxel words), NULL when they store centred rows. */
int lrg_front_step(LrgSlot *slots, LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                   const LrgWeights *weights, const LrgPackedBuffers *buffers, void *stream);
const float *lrg_packed_rows_center(const LrgGrowParams *params, const LrgPackedBuffers *buffers);

/* `iterations` calls of lrg_grow_step_packed captured into a HIP graph on `stream` (not the null stream; weights->packed
 * set).  Nothing runs at creation.  lrg_step_graph_launch replays them with one host call. */
int lrg_step_graph_create(LrgSlot *slots, LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                          const LrgWeights *weights, const LrgPackedBuffers *buffers, int iterations, void *stream,
                          void **graph_out);
int lrg_step_graph_launch(void *graph, void *stream);
int lrg_step_graph_destroy(void *graph);

/* ------------------------------------------------------------------------------------------------
 * Free-running iterations: the same loop (test_region_grow.py:208-306), every slot at its own pace inside ONE launch.
 * lrg_grow_step_packed is five launches over all slots, each as long as its slowest slot or tile; rooms are independent
 * (:110-183), so here a slot's stages -- front (mask update :262-288, stop decision :291-306, commit / next seed :186-217, box
 * query :221-235, medians :241, sampling :237-252, gather :242-254), branch stacks, pooled product, head stacks
 * (learn_region_grow_util.py:106-162) -- are ordered by that slot's own arrival counters: front workgroups serve the slots,
 * the other CUs run tile tasks from a queue (csrc/lrg_async.inl).  Greedy growing (restarts = group_size = 1) on rooms with
 * packed voxel words, n_inlier / n_neighbor <= 512, lite 0 / 2 -- LRG_EINVAL otherwise (use lrg_grow_step_packed).
 * State between calls is exactly that of lrg_grow_step_packed (a call ends every slot between two evaluations, logits in
 * place), so the two may alternate on the same buffers; results are identical bit for bit.
 * ---------------------------------------------------------------------------------------------- */
#define LRG_REG_TILE_AUTO_MIN 1    /* register-tile launches (branch_waves = 1) by default from this many (up to 24 slots a branch tile is two tasks: one room +14 %, x 3 regions +7 %, 16 rooms +3 %) ... */
#define LRG_REG_TILE_AUTO_MAX 176  /* ... to this many slots in flight: 16 (x 3 regions) / 34 / 68 / 100 / 136 / 160 slots +6 / +11.5 / +9.6 / +9.1 / +4 / +4.6 % over the one-kernel launch,
                                      192: -1.7 % (two teams per CU are too few there), eight 100 k-point scenes x 3 regions: -0.7 % (profiles/r06_reg_tiles_slots.txt) */
typedef struct LrgAsyncBuffers {
    int32_t *queue;             /* lrg_grow_async_queue_bytes(n_slots) bytes, 256-byte aligned: task ring + control words (cleared by every call) */
    size_t queue_bytes;
    int32_t *sync;              /* [n_slots, 16] arrival counters of the slots (cleared by every call)                     */
    int32_t front_workgroups;   /* workgroups serving the slots (each up to 8 of them), 0 = default                       */
    int32_t teams;              /* tile teams (four wavefronts) per worker workgroup: 1 .. 4 (4: two of them run branch tiles only), 0 = default */
    int32_t compute_units;      /* workgroups of the launch in all (front + worker), at most one per CU of the device; 0 = all CUs */
    int32_t poll_sleep;         /* idle tile teams poll the queue every poll_sleep x ~0.25 us; 0 = default                  */
    int32_t branch_parts;       /* tasks per branch tile (1, 2 or 4: they share the four column blocks of its pooled layer and each run the
                                   layers before it again); 0 = by the number of slots (2 up to 12 slots, else 1)                    */
    int32_t gemv_units;         /* 0 = default: up to 176 slots the heads' pooled kernels stay in the LDS of 2 C / 32 workgroups of their own (32
                                   columns each; a slot's pooled product = one 4 KB row in, 32 sums out per unit, and its head tiles start
                                   beside it) where that fits; 1 = also above 176 slots; -1 = the tile teams compute it in 128-column
                                   blocks from L2                                                                                 */
    int32_t *room_queue;        /* nullable: rooms waiting for a slot -- [0] rooms handed out so far (the caller zeroes it when it refills
                                   the queue), [1] rooms queued, [2 + k] = room index | reset << 30 (reset: clear visited / labels /
                                   cursor first, as lrg_bind_group does).  A slot whose room is finished (or that has none) takes the
                                   next one inside the launch; finished rooms appear in the stats ring as slot | room << 32      */
    uint64_t *work;             /* nullable: [8] running totals (never cleared by the library) of what the launches evaluated: LrgNet
                                   evaluations, distinct inlier rows, distinct neighbour rows, 32-row tiles per stack (branch = head) --
                                   the algorithmic FLOPs of the launches follow from these; [4] regions voided and grown again under
                                   `speculate`, [5] the evaluations those regions had taken, [6] of these: steps counted in stats[2]
                                   (ABI 9: 8 words, was 4)                                                                          */
    /* In-launch fill-in (ABI 8; all five non-NULL, 13 features): a room that finishes during the launch gets its 1-NN fill-in
       (test_region_grow.py:308-316) from tile teams of the same launch instead of from lrg_nn1_fill_batch between launches; such rooms
       carry bit 31 in the slot word of their done-ring entry.  The arenas are laid out like the label arena the rooms' LrgRoom.label
       pointers point into (fill_label_base = its first element): a room's lists live at its label offset.                              */
    int32_t *fill_list;         /* [points of all rooms] int32                                                                          */
    uint64_t *fill_best;        /* [points of all rooms] 8-byte aligned                                                                  */
    int32_t *fill_sync;         /* [fill_rooms, 4] int32                                                                                 */
    const int32_t *fill_label_base;
    int32_t *fill_out_base;     /* the filled labels (label_out of lrg_nn1_fill), same layout                                            */
    int32_t fill_rooms;         /* rooms in the LrgRoom array                                                                            */
    int32_t fill_wgs;           /* worker workgroups with a team for the fill-in ring (one more than the others have, up to three tile teams;
                                   else their last team); 0 = default (64)                                                                */
    int32_t rows16;             /* 1: buffers->x_in / x_nb hold row_cap x 16 floats (16-byte aligned): lrg_grow_async gathers its rows at a 64-byte stride
                                   in 16-byte pieces (9 .. 16 features); 0: row_cap x feature_size floats, one element per store (ABI 8)            */
    int32_t speculate;          /* K > 1 (ABI 9): the slots are groups of K (slots g K .. g K + K - 1, one front workgroup per group), all bound to the SAME room:
                                   the regions of the room's next K unvisited seeds grow side by side and are committed in seed order; a region is dropped and
                                   grown again when a region committed before it contains a point inside any dilated box it has queried -- otherwise its
                                   queries saw exactly what the sequential loop (:186-188,:210-217,:227-228) would have shown them.  Same regions, same labels;
                                   for the few-rooms corner (one room or scene per GPU), where a room's chain of dependent steps leaves the chip idle.
                                   n_slots must be a multiple of K; the caller binds whole groups (lrg_bind_group with group_size K); 0 / 1 = off           */
    int32_t branch_waves;       /* (ABI 10; was `reserved2`) two-kernel launches: the CUs that run tiles as a SECOND kernel (512 threads, up to 256 VGPRs) resident beside the
                                   front workgroups' and units' kernel (csrc/lrg_wave_tile.inl; same results bit for bit in every form):
                                     1 = REGISTER TILES: a branch tile (learn_region_grow_util.py:106-123) by a team of four wavefronts that each run layers 0 - 2 in
                                         registers, meet once in LDS behind layer 3 and take a quarter of the pooled layer each; the other team of the CU runs head tiles;
                                     4 / 8 = one-wavefront tasks: a PREFIX task (layers 0 - 3) and four POOL tasks (a quarter of the pooled layer) per tile, on CUs that
                                         hold the kernels of their stage in LDS, that many wavefronts per such CU (measured slower at every slot count: kept as an option);
                                     0 = by the slot count: register tiles from LRG_REG_TILE_AUTO_MIN to LRG_REG_TILE_AUTO_MAX slots; -1 = one kernel, team tiles.
                                   Needs the paper's network (13 -> 64 -> 64 -> 64 -> 128 -> 512), rows16, no tail_ctl / pool_rows, all CUs of the device. */
    int32_t start_wait_us;      /* the launch's start rendezvous (all its workgroups must be running at once): how long the front workgroups wait for the
                                   others before the launch gives up with reason 6; 0 = default: the launch budget + 20 ms (a kernel of another stream that
                                   holds CUs for up to one budget is waited out, something that never leaves is reported)  (ABI 9; was `reserved`) */
    float *pool_rows;           /* nullable: lrg_grow_async_pool_rows_bytes(weights, n_slots) bytes, 16-byte aligned -- with the pooled-product units a
                                   branch tile leaves the column maxima of its rows as one row here (16-byte stores) and the units take the maximum
                                   over a slot's tiles; NULL: one atomicMax per column and tile on the pooled feature (ABI 8)              */
    size_t pool_rows_bytes;
    uint64_t *debug_ticks;      /* nullable: [64] accumulators (never cleared by the library) of wall-clock ticks by stage of the
                                   launch, for tools/free_run_perf.py (layout: csrc/lrg_async.inl, LrgAsyncArgs.dbg)            */
    /* Shared tail tiles (ABI 9; tail_ctl non-NULL, tail_rows > 0, rows16): a slot's rows beyond its last full 32-row tile (16 of ~91 rows per side on
       average: as padded tiles of the slot's own they were 18 % of all tile rows) go to `tail_rows` rows per side that ALL slots share -- rows
       [n_slots * row_stride, + tail_rows) of the row arrays (buffers->row_cap and the workspace must cover them and 32 rows more) -- so that the tails of several slots fill one
       branch tile.  Same results bit for bit (the shared tile is lrg_forward_packed's tile: runs of rows, per-run max-pool).  When a launch runs out of
       shared rows the slots pad tiles of their own again.                                                                                        */
    int32_t *tail_ctl;          /* lrg_grow_async_tail_bytes(n_slots, tail_rows) bytes, 64-byte aligned (cleared by every call)                    */
    int32_t tail_rows;          /* a multiple of 32                                                                                                */
    int32_t tail_close_us;      /* a slot closes its last shared tile (the rest of its rows stay empty) when nobody has filled it up that long after the
                                   slot's rows went out; 0 = default (2 us), -1 = at once                                                          */
} LrgAsyncBuffers;

size_t lrg_grow_async_queue_bytes(int n_slots);
size_t lrg_grow_async_tail_bytes(int n_slots, int tail_rows);
size_t lrg_grow_async_pool_rows_bytes(const LrgWeights *weights, int n_slots);

/* One free-running launch: every slot takes up to max_steps evaluations (grow steps), and starts no new one once budget_us
 * microseconds have passed since the launch began (0 = no time limit); slots whose room is finished or that are not bound
 * leave at once.  `buffers` are those of lrg_grow_step_packed with row_cap >= n_slots * 32 * ceil(max(n_inlier, n_neighbor) / 32)
 * (slot s owns that many rows from s * that number on).
 * Residency: the launch is one workgroup per CU whose roles wait for each other, so all of them must run at once.  Refused with
 * LRG_ERESIDENCY, before anything is enqueued, when the kernel does not fit a CU or the stream's CU mask (hipExtStreamCreateWithCUMask)
 * leaves it fewer CUs than the launch has workgroups (AsyncBuffers.compute_units: pass the number of CUs the stream may use).  CUs held
 * by somebody else (another process, a kernel of another stream) cannot be seen from the host: every workgroup reports in when it
 * starts, and a launch whose workgroups have not all started within 20 ms gives up at once (reason 6 in stats[3]) instead of
 * stalling.  A hand-over that was given up is counted in stats[3] (low word: workgroups that left this way, high word: sum of
 * their reasons -- 1 launch past its time limit, 2 a team waited too long for a task, 3 a team lost a wavefront at a barrier, 4 / 5 a
 * pooled-product unit / a head tile waited too long, 6 not all workgroups resident); the caller must treat a non-zero count as a
 * failed call. */
int lrg_grow_async(LrgSlot *slots, LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                   const LrgWeights *weights, const LrgPackedBuffers *buffers, const LrgAsyncBuffers *async_buffers,
                   int max_steps, int budget_us, void *stream);

/* A stream whose kernels run only on the compute units set in `mask` (bit i of word i / 32; hipExtStreamCreateWithCUMask).
 * Meant for lanes (slot groups iterating independently on their own streams, the batched scheduler of north_star) on disjoint CU
 * sets.  mask = NULL: a plain non-blocking stream.
 * Measured on one MI355X: no gain at two lanes (lanes do not compete for CUs, profiles/r02_lanes_cu_partition_sweep.txt), and three
 * masked streams of a spin-kernel microbenchmark did not finish (tools/stream_overlap.hip) -- an option for experiments, off by default. */
int lrg_stream_create_cu_mask(const uint32_t *mask, int words, void **stream);
int lrg_stream_destroy(void *stream);

/* ------------------------------------------------------------------------------------------------
 * Beam search (test_beam_search.py:143-290) with the queue on the device.
 * A room in flight is a group of beam_width * search_width consecutive slots (child (qid, search id) = slot qid * search_width
 * + search id).  lrg_beam_level = one level of every room in flight: lrg_beam_advance (children of the last level -> next queue
 * by size, :275-287; stall test / commit of the head, :188-198,:289-293; next seed, :154-174; next children into the slots),
 * then the loop's kernels over all slots (box query, median, sampling, gather, LrgNet on the distinct rows, mask update, scan of
 * the updated masks).  No host decision in between; finished rooms appear in the stats ring (first slot of the group).
 * ---------------------------------------------------------------------------------------------- */
typedef struct LrgBeamGroup {
    uint8_t *parent;         /* [beam_width, cap] masks of the queue entries                                             */
    int32_t cap;             /* row stride of parent                                                                     */
    int32_t room;            /* index into rooms[], -1 = none (the host binds a group by writing room, seed = -1, pending = done = 0) */
    int32_t seed;            /* seed point in progress, -1 = pick the next                                               */
    int32_t level, stuck, steps, nq;
    int32_t pending;         /* 1: the slots hold the evaluated children of the current queue                            */
    int32_t done;            /* the room has no unvisited seed left                                                      */
    int32_t seq_mn[3], seq_mx[3];
    int32_t q_count[16], q_parent[16];       /* queue entries: mask size, parent buffer (-1 = the seed alone)            */
    int32_t q_mn[16][3], q_mx[16][3];        /* their bounding boxes                                                      */
    int32_t pad;
} LrgBeamGroup;
int lrg_beam_advance(LrgBeamGroup *groups, LrgSlot *slots, LrgRoom *rooms, int n_groups, int beam_width, int search_width,
                     const LrgGrowParams *params, int64_t *stats, void *stream);
int lrg_beam_level(LrgBeamGroup *groups, LrgSlot *slots, LrgRoom *rooms, int n_groups, int beam_width, int search_width, int max_points,
                   const LrgGrowParams *params, const LrgWeights *weights, const LrgStepBuffers *buffers, unsigned forward_flags,
                   void *stream);

/* 1-NN fill-in of unlabeled points in all F feature dims, first-min ties (:308-316). */
int lrg_nn1_fill(const float *points, int n, int F, const int32_t *label_in, int32_t *label_out, void *stream);
/* The same result through a tiled search that reads the candidate rows once per 64 unlabeled points.  workspace: lrg_nn1_fill_workspace_bytes(n) bytes of device memory, 256-byte aligned.  F <= 13.  (A NaN distance
 * sorts after every number here; numpy.argmin would return the first NaN.) */
size_t lrg_nn1_fill_workspace_bytes(int n);
int lrg_nn1_fill_ws(const float *points, int n, int F, const int32_t *label_in, int32_t *label_out, void *workspace,
                    size_t workspace_bytes, void *stream);
/* Several rooms at once (the rooms that finish during one free-running launch): the same result as lrg_nn1_fill_ws room by room, three
 * launches per 64 rooms.  `jobs` is host memory; n = 0 rooms are skipped.  workspace: lrg_nn1_fill_batch_workspace_bytes(jobs, n_jobs) bytes. */
typedef struct LrgFillJob {
    const float *points;         /* [n, F] */
    const int32_t *label_in;     /* [n], 0 = unlabeled */
    int32_t *label_out;          /* [n] */
    int32_t n;
    int32_t reserved;
} LrgFillJob;
size_t lrg_nn1_fill_batch_workspace_bytes(const LrgFillJob *jobs, int n_jobs);
int lrg_nn1_fill_batch(const LrgFillJob *jobs, int n_jobs, int F, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * tf_ops/grouping replacements.  Same argument order as the reference launchers
 * (tf_ops/grouping/tf_grouping_g.cu:125-141) plus a trailing stream; extern "C"; int return.
 * ---------------------------------------------------------------------------------------------- */
/* queryBallPointLauncher (tf_grouping_g.cu:125; kernel :3-36).  Rows with no hit are zero-filled. */
int lrg_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                         int *idx, int *pts_cnt, void *stream);
/* query_ball_point + group_point(xyz, idx) + group_point(points, idx) as the reference chains them in sample_and_group (train_pointnet.py:113-121), in ONE launch:
 * idx [b,m,nsample], pts_cnt [b,m] as lrg_query_ball_point; grouped_xyz [b,m,nsample,3] = xyz1 rows at idx (minus the query's coordinates when subtract_center: the
 * float32 subtraction of :117); grouped_points [b,m,nsample,c] = points rows at idx (points and grouped_points both NULL: skipped).  nsample <= 4096.  (ABI 9) */
int lrg_query_ball_group(int b, int n, int m, int c, float radius, int nsample, const float *xyz1, const float *xyz2, const float *points, int *idx,
                         int *pts_cnt, float *grouped_xyz, float *grouped_points, int subtract_center, void *stream);

/* selectionSortLauncher (tf_grouping_g.cu:129; kernel :83-123): full [b,m,n] outputs, first k sorted. */
int lrg_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream);
/* groupPointLauncher (tf_grouping_g.cu:133; kernel :40-57) */
int lrg_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out,
                    void *stream);
/* groupPointGradLauncher (tf_grouping_g.cu:137; kernel :61-78); grad_points must be zeroed by the caller
 * (the reference op does cudaMemset at tf_grouping.cpp:204). */
int lrg_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                         float *grad_points, void *stream);
/* knn_point (tf_grouping.py:48-73) in one kernel: val [b,m,k], idx [b,m,k] = the first k entries of selectionSortLauncher applied
 * to the distance matrix sum_c (xyz1[b,n,c]-xyz2[b,m,c])^2, bit for bit (same swap sequence, so the same order among equal
 * distances), without the b x m x n matrix ever existing.  n <= 4096 (a row lives in registers), k <= min(n, 512); LRG_EINVAL - 2 beyond. */
int lrg_knn_topk(int b, int n, int m, int c, int k, const float *xyz1, const float *xyz2, float *val, int *idx, void *stream);
/* dist[b,m,n] = sum_c (xyz1[b,n,c]-xyz2[b,m,c])^2 -- the matrix knn_point builds at tf_grouping.py:62-65 */
int lrg_pairwise_sqdist(int b, int n, int m, int c, const float *xyz1, const float *xyz2, float *dist, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Training side: the pieces of the backward pass and of AdamOptimizer
 * (learn_region_grow_util.py:165-189) that the forward entry points do not cover.  The step is sequenced by the host
 * mirror (learn_region_grow_amd/train.py: LrgNetTrainer.train_step = sess.run([net.train_op, net.loss, ...]) at
 * train_region_grow.py:175).
 * ---------------------------------------------------------------------------------------------- */
/* C[M,N] = op(A)[M,K] op(B)[K,N] in fp32 on the matrix cores.  transA: A is stored [K,M] (row stride lda), else [M,K];
 * transB: B is stored [N,K] (row stride ldb), else [K,N].  Epilogue: C = acc + addend (nullable, [M,N] with row stride ldc),
 * then C = 0 where mask <= 0 (nullable, same shape: the ReLU gradient of the layer whose output the result is the gradient of).
 * split_k > 1: the reduction is split over that many workgroup layers which ADD into C -- the caller zeroes C first, and no
 * epilogue is allowed.  dX = dZ W^T is (transB = 1); dW = X^T dZ over all rows is (transA = 1, split_k ~ rows / 1024). */
int lrg_gemm_f32(int M, int N, int K, const float *A, int lda, int transA, const float *B, int ldb, int transB, float *C, int ldc,
                 const float *addend, const float *mask, int split_k, void *stream);
/* d(loss)/d(logits) of a sparse softmax cross entropy over `rows` two-class slots with per-class weights
 * (add head :174: w_pos = w_neg = 1 / rows; remove head :166-172: 1 / #positive and 1 / #negative slots of the batch).
 * stats (device double[8], ACCUMULATED): 0 weighted loss, 1 slots with argmax == label, 2 true positives, 3 predicted
 * positives, 4 labelled positives, 5 rows (the numerators of add_acc, add_prc, add_rcl ... at :175-184). */
int lrg_ce_grad(const float *logits, const int32_t *labels, long rows, float w_pos, float w_neg, float *dlogits, double *stats,
                void *stream);
/* Gradient of the column max over each instance's rows (:122-123) followed by the pooled layer's ReLU: y [B,rows,C] is that
 * layer's output, dpool [B, dpool_stride] the gradient of its column maxima; dy [B,rows,C] gets dpool / (number of rows that tie
 * for the maximum) on those rows -- tf.reduce_max's rule -- and 0 elsewhere (and 0 everywhere when the maximum is 0). */
int lrg_pool_backward(const float *y, const float *dpool, int B, int rows, int C, int dpool_stride, float *dy, void *stream);
/* out[s, n] = sum of x[s*seg_rows + r, n] over r < seg_rows, for s < n_seg (bias gradients, the gradient of the tiled pooled
 * feature: the sum over an instance's rows). */
int lrg_segment_colsum(const float *x, long n_seg, int seg_rows, int N, float *out, void *stream);
/* One AdamOptimizer update of n parameters as TensorFlow 1 applies it (:188): m += (g - m)(1 - beta1); v += (g^2 - v)(1 - beta2);
 * p -= lr_t m / (sqrt(v) + epsilon), with lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t) computed by the caller. */
int lrg_adam_step(float *params, const float *grads, float *m, float *v, long n, float lr_t, float beta1, float beta2, float epsilon,
                  void *stream);

/* ------------------------------------------------------------------------------------------------
 * Room preprocessing P0 (replaces the per-room block test_region_grow.py:119-173: equalisation, per-point PCA normals and
 * curvature, feature stack).  SURVEY.md section 8f, row f1.
 *
 * raw [n_raw, raw_stride >= 6] float32 rows (x y z r g b ...), obj_id / cls_id [n_raw] (nullable).  Outputs (device,
 * capacity n_raw rows): equalized_idx [N] = raw index of the first point of every resolution-sized voxel, in file order
 * (:127-129,:134); unequalized_idx [n_raw] = equalised row of each raw point (:130); *n_equalized (device int32) = N.
 * eig_mode 0: cov [N,9] float64 = the covariance of :157, bit-identical to the NumPy loop (same neighbour and summation
 *             order); the host finishes with numpy.linalg.svd exactly as the reference does.  points may be NULL.
 * eig_mode 1: also points [N,feature_size] float32 (:165-172), obj_out / cls_out [N], curvatures [N] float64 (:163), with the
 *             3x3 decompositions done here by Jacobi iteration in float64 (agrees with LAPACK to ~1e-16 |cov|).
 * eig_mode 2: eig_mode 1 prepared for an exact finish on the host (learn_region_grow_amd.preprocess_gpu, eig='exact'): cov is written too,
 *             curvatures stay UN-normalised (S[2]/sum(S), :160-161 -- the division by the maximum, :163, is the host's, with LAPACK's own
 *             maximum), and lrg_preprocess_unsafe_normals tells which points' float32 normals are not certain to equal what
 *             numpy.linalg.svd would give (near-degenerate eigenvalue pairs, components next to a float32 rounding boundary): the host
 *             redoes those few with LAPACK and gets the reference's features and seed order bit for bit at the all-GPU rate.
 * workspace: lrg_preprocess_workspace_bytes(n_raw) bytes, 256-byte aligned.  Points whose voxel falls outside a 21-bit
 * window per axis are reported by lrg_preprocess_status (1) and left out. */
size_t lrg_preprocess_workspace_bytes(int n_raw);
int lrg_preprocess(const float *raw, int raw_stride, const int32_t *obj_id, const int32_t *cls_id, int n_raw, float resolution,
                   int feature_size, int eig_mode, void *workspace, size_t workspace_bytes, float *points, int32_t *obj_out,
                   int32_t *cls_out, double *curvatures, int32_t *equalized_idx, int32_t *unequalized_idx, double *cov,
                   int32_t *n_equalized, void *stream);
int lrg_preprocess_status(const void *workspace, int n_raw, int32_t *host_status, void *stream);
/* after lrg_preprocess(..., eig_mode 2, ...) on this workspace: flags_out [n_equalized] int32 (device), 1 = redo this point's decomposition with LAPACK */
int lrg_preprocess_unsafe_normals(const void *workspace, int n_raw, int n_equalized, int32_t *flags_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LRG_HIP_H */
