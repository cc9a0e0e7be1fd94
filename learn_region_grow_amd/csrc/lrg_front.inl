// Packed-row formulation of one lock-step iteration (lrg_grow_step_packed); included by lrg_grow.hip.
//
// lrg_grow_step is a chain of nine launches whose durations are latency, not work (profiles/r02_diag_gaps.txt: advance
// 24 us, box count 7, box compact 8, medians 15, prepare 15, branch stack 54, pooled GEMM 10, head stack 26, mask update
// 7), and its network kernels spend ~45 % of their 32-row tiles on padding: every slot's stacked set is tiled on its own,
// and the median region has 57 points.  Here
//   * everything between two network evaluations is ONE kernel with one 1024-thread workgroup per slot
//     (lrg_front_kernel): the mask update of the evaluation just finished (test_region_grow.py:262-288), count and
//     bounding box of the new mask (:292-293) from the slot's index lists instead of a scan of the room, the stop / stuck
//     decision (:291-306), commit + next seed (:186-217), the dilated box query with ordered compaction (:221-235) as
//     one coalesced pass over the room with all flags parked in LDS, the per-channel medians (:241), subset sampling
//     (:237-252) and the gather (:242-254);
//   * the gather writes only the DISTINCT rows of each stacked set (the rest are copies, :240,:252), and writes them
//     back to back for all slots: the network runs on dense 32-row tiles of that packed array (lrg_forward_packed), with
//     the max-pool (:122-123) and the hoisted per-instance bias (:128-141) applied per run of rows of one slot.
// Four launches per iteration (front | branch stacks | pooled GEMM | head stacks); results are bit-identical to
// lrg_grow_step (same arithmetic on the same rows; the max-pool and the atomics are order-independent).

#define LRG_FRONT_THREADS 1024
#ifndef LRG_ASYNC_DEBUG
#define LRG_ASYNC_DEBUG 0
#endif
#define LRG_FRONT_MAXCHUNK 32                         // 32 x 4096 points: rooms up to 131072 points (KITTI scenes: ~100 k)
#define LRG_FRONT_MAXSAMPLE 1024                      // n_inlier, n_neighbor <= 1024

struct LrgFrontArgs {
    float *center;
    int32_t *sample_in, *sample_nb;
    float *x_in, *x_nb;
    int32_t *row_slot_in, *row_slot_nb;
    float4 *upd_in, *upd_nb; // [n_slots, n_inlier] / [n_slots, n_neighbor]: (x, y, z as stored in the packed row, ground-truth flag) of the
                             // slot's distinct rows -- what the NEXT mask update needs of them, in storage of the slot's own.  The packed
                             // arrays are re-allocated from row 0 by every launch: a workgroup that starts late (the chip busy with
                             // another lane's kernels) would find its rows of the last iteration overwritten by the slots that are
                             // already gathering.
    const float *rmv_logits, *add_logits;
    int32_t *slot_rows;      // [n_slots,4]: rows_in, rows_nb, first packed inlier row, first packed neighbour row
    int32_t *counters;       // [0] packed inlier rows, [1] packed neighbour rows allocated so far in this iteration
    float *pooled;           // [n_slots, pooled_stride] pooled features of the network workspace (zeroed here per slot)
    int pooled_stride;
    int64_t *stats;
    int64_t *phase_ticks;    // nullable: [n_slots,2] wall-clock ticks per slot: (0) update / stop / commit, (1) query / median / gather
    int own_medians;         // greedy front kernel: 1 = every slot's workgroup computes its nine medians itself (no launch of their own)
    unsigned long long *phase_dbg;   // nullable (free-running kernel): [8] accumulated wall-clock ticks of the front's phases
    int row_stride;          // free-running kernel: slot s owns the rows [s * row_stride, (s + 1) * row_stride) of the row arrays
    int rows16;              // free-running kernel: 1 = the gathered rows are written at a 64-byte stride (16 floats, zero-padded) in 16-byte pieces
    int fill_in_launch;      // free-running kernel: finished rooms are filled in (:308-316) by tile teams of the same launch -- flagged in the done ring (bit 31 of the slot word)
    // free-running kernel, shared tail tiles (lrg_async.inl): a slot's rows beyond its last FULL 32-row tile go to rows that the slots share, reserved from a cursor
    // per side -- several slots' tails fill one tile instead of each padding a tile of its own.  nullable (then every slot pads its own tail, as before).
    int32_t *tail_cur;       // [0] / [16]: rows reserved so far on the inlier / neighbour side (one 64-byte line each)
    int32_t *tail_base;      // [n_slots][2]: where the slot's tail rows of its evaluation in flight start in the shared rows (-1: in its own place)
    int tail_rows;           // shared rows per side (a multiple of 32)
    int tail_row0;           // first shared row in the row arrays (= n_slots * row_stride)
    unsigned long long *spec_stats;      // nullable: [0] regions voided by an earlier commit, [1] evaluations those regions had taken, [2] of them: mask updates done, i.e. steps the
                                         // device's step counter holds that no committed region keeps (LrgAsyncBuffers.work + 4)
    int spec_k;              // free-running kernel: K > 1 = speculation -- the slots g K .. g K + K - 1 grow the regions of the next K unvisited seeds of ONE room
                             // side by side (LrgAsyncBuffers.speculate; see "speculation" below); 0 / 1 = one slot, one room
};

// ---- (1) mask update of the evaluation just finished + count / bounding box of the new mask + stop decision ----
// The new mask is (old members that survive the removes) + (newly added points that survive them): both sets are at
// hand as index lists -- cur_idx (written by the box query that preceded the evaluation) and the list of points this
// update switched on -- so :292-293 cost O(region), not O(room).
__device__ void lrg_front_update(LrgSlot *S, const LrgRoom *R, int s, const LrgGrowParams &prm, const LrgFrontArgs &a,
                                 int *sh_added, int *red) {
    __shared__ int sh_upd, sh_nadd, sh_acc[2];
    const int tid = threadIdx.x, bd = blockDim.x;
    const int F = prm.feature_size, Ni = prm.n_inlier, Nn = prm.n_neighbor;
    const int nc = S->nc, ne = S->ne;
    const int offi = a.slot_rows[4 * s + 2], offn = a.slot_rows[4 * s + 3];
    const float res = prm.resolution;
    const float c0 = a.center[s * 16 + 0], c1 = a.center[s * 16 + 1];
    const uint32_t k0 = prm.rng_seed, k1 = (uint32_t)R->room_id;
    const uint32_t seed = (uint32_t)S->seed, restart = (uint32_t)S->restart, step = (uint32_t)S->step;
    uint8_t *cur = S->cur;
    if (tid == 0) { sh_upd = 0; sh_nadd = 0; sh_acc[0] = 0; sh_acc[1] = 0; }
    __syncthreads();
    // ---- add pass (:266,:270-273,:283-285): sample slot j draws for itself, its logits / coordinates are its source row's ----
    for (int j = tid; j < Nn; j += bd) {
        const int srow = ne < Nn ? a.sample_nb[(long)s * Nn + j] : j;
        const long row = offn + srow;
        const float4 u = a.upd_nb[(long)s * Nn + srow];
        if ((a.add_logits[2 * row + 1] > a.add_logits[2 * row] ? 1 : 0) == (u.w != 0.f ? 1 : 0)) atomicAdd(&sh_acc[0], 1);   // add_acc
        bool take;
        if (prm.policy == 2) take = u.w != 0.f;
        else {
            const float conf = lrg_conf(a.add_logits + 2 * row);
            if (prm.policy == 1) take = conf > 0.5f;
            else take = lrg_uniform01(lrg_rng_word((uint32_t)j, LRG_PURPOSE_ADD, seed, restart, step, k0, k1)) < conf;
        }
        if (take) {
            const int vx = lrg_voxel_of(__fadd_rn(u.x, c0), res);       // :271-272: un-centre x,y then rint(/res)
            const int vy = lrg_voxel_of(__fadd_rn(u.y, c1), res);
            const int vz = lrg_voxel_of(u.z, res);
            const int idx = lrg_hash_lookup(R->hash_keys, R->hash_vals, R->hash_mask, lrg_pack_voxel(vx, vy, vz));
            if (idx >= 0) {
                // several sample slots may name the same point: the word-wide atomic elects the one that switches it on
                unsigned *w = reinterpret_cast<unsigned *>(cur + (idx & ~3));
                const unsigned bit = 1u << (8 * (idx & 3));
                if (!(atomicOr(w, bit) & bit)) { sh_upd = 1; sh_added[atomicAdd(&sh_nadd, 1)] = idx; }
            }
        }
    }
    __syncthreads();
    // ---- remove pass (:267,:274-277,:286-287) ----
    for (int j = tid; j < Ni; j += bd) {
        const int srow = nc < Ni ? a.sample_in[(long)s * Ni + j] : j;
        const long row = offi + srow;
        const float4 u = a.upd_in[(long)s * Ni + srow];
        if ((a.rmv_logits[2 * row + 1] > a.rmv_logits[2 * row] ? 1 : 0) == (u.w != 0.f ? 1 : 0)) atomicAdd(&sh_acc[1], 1);   // remove_acc
        bool take;
        if (prm.policy == 2) take = u.w != 0.f;
        else {
            const float conf = lrg_conf(a.rmv_logits + 2 * row);
            if (prm.policy == 1) take = conf > 0.5f;
            else take = lrg_uniform01(lrg_rng_word((uint32_t)j, LRG_PURPOSE_RMV, seed, restart, step, k0, k1)) < conf;
        }
        if (take) {
            const int vx = lrg_voxel_of(__fadd_rn(u.x, c0), res);
            const int vy = lrg_voxel_of(__fadd_rn(u.y, c1), res);
            const int vz = lrg_voxel_of(u.z, res);
            const int idx = lrg_hash_lookup(R->hash_keys, R->hash_vals, R->hash_mask, lrg_pack_voxel(vx, vy, vz));
            if (idx >= 0) cur[idx] = 0;
        }
    }
    __syncthreads();
    // ---- members and bounding box of the new mask (:292-293) from the two lists ----
    const int32_t *vox = R->voxels;
    const int32_t *lst = S->cur_idx;
    const int nadd = sh_nadd;
    int cnt = 0;
    int mn0 = INT_MAX, mn1 = INT_MAX, mn2 = INT_MAX, mx0 = INT_MIN, mx1 = INT_MIN, mx2 = INT_MIN;
    for (int i0 = 0; i0 < nc; i0 += 4 * bd) {
        int id[4], al[4], va[4], vb[4], vc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) id[k] = lst[min(i0 + k * bd + tid, nc - 1)];       // unconditional, clamped: all in flight
#pragma unroll
        for (int k = 0; k < 4; ++k) { al[k] = cur[id[k]]; va[k] = vox[3 * id[k]]; vb[k] = vox[3 * id[k] + 1]; vc[k] = vox[3 * id[k] + 2]; }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k * bd + tid < nc && al[k]) {
                ++cnt;
                mn0 = min(mn0, va[k]); mn1 = min(mn1, vb[k]); mn2 = min(mn2, vc[k]);
                mx0 = max(mx0, va[k]); mx1 = max(mx1, vb[k]); mx2 = max(mx2, vc[k]);
            }
    }
    for (int k = tid; k < nadd; k += bd) {
        const int idx = sh_added[k];
        if (cur[idx]) {                                               // (a point added and removed in the same step stays out)
            const int va = vox[3 * idx], vb = vox[3 * idx + 1], vc = vox[3 * idx + 2];
            ++cnt;
            mn0 = min(mn0, va); mn1 = min(mn1, vb); mn2 = min(mn2, vc);
            mx0 = max(mx0, va); mx1 = max(mx1, vb); mx2 = max(mx2, vc);
        }
    }
    lrg_block_bbox(cnt, mn0, mn1, mn2, mx0, mx1, mx2, red);
    if (tid == 0) {
        S->scan_cnt = cnt;
        S->scan_mn[0] = mn0; S->scan_mn[1] = mn1; S->scan_mn[2] = mn2;
        S->scan_mx[0] = mx0; S->scan_mx[1] = mx1; S->scan_mx[2] = mx2;
        S->updated = sh_upd;
        S->pad = 0;
        S->step += 1;
        S->steps_total += 1;                                                    // :288
        S->acc_add = sh_acc[0]; S->acc_rmv = sh_acc[1];
        if (a.stats) atomicAdd(reinterpret_cast<unsigned long long *>(&a.stats[2]), 1ULL);
        lrg_stop_logic(S);                                                      // :291-306
    }
    __syncthreads();
}

// ---- --scoring ml (test_random_restart.py:251-271): log-likelihood of the masks this step sampled ----
// Every sample slot i contributes log(conf_i) / 512 if its (un-centred, re-voxelised) point lies in the set of voxels whose
// draw fired, else log(1 - conf_i) / 512 -- membership is by VOXEL, so a copy of an added point counts as added whatever its own
// draw said.  float32 terms as NumPy computes them, summed in double in a fixed order (deterministic).  Both sides divide by
// NUM_NEIGHBOR_POINT (:261,:269).  Accumulated per restart in slot.ml_score (upstream's reset to a list at :194-196 breaks the
// accumulation from the second restart on; this is the scalar accumulator it evidently meant).
__device__ void lrg_front_ml_score(LrgSlot *S, const LrgRoom *R, int s, const LrgGrowParams &prm, const LrgFrontArgs &a) {
    __shared__ unsigned long long sh_key[LRG_FRONT_MAXSAMPLE];
    __shared__ double sh_part[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int F = prm.feature_size;
    const float c0 = a.center[s * 16 + 0], c1 = a.center[s * 16 + 1];
    const uint32_t k0 = prm.rng_seed, k1 = (uint32_t)R->room_id;
    const uint32_t seed = (uint32_t)S->seed, restart = (uint32_t)S->restart, step = (uint32_t)S->step;
    double total = 0.0;
    for (int side = 0; side < 2; ++side) {                      // 0: add head on the neighbour slots, 1: remove head on the inlier slots
        const int N = side ? prm.n_inlier : prm.n_neighbor, nside = side ? S->nc : S->ne;
        const int off = a.slot_rows[4 * s + (side ? 2 : 3)];
        const bool have = tid < N;
        unsigned long long key = LRG_HASH_EMPTY;
        float conf = 0.f;
        bool take = false;
        if (have) {
            const int srow = nside < N ? (side ? a.sample_in : a.sample_nb)[(long)s * N + tid] : tid;
            const long row = off + srow;
            const float4 u = (side ? a.upd_in : a.upd_nb)[(long)s * N + srow];
            conf = lrg_conf((side ? a.rmv_logits : a.add_logits) + 2 * row);
            if (prm.policy == 2) take = u.w != 0.f;
            else if (prm.policy == 1) take = conf > 0.5f;
            else take = lrg_uniform01(lrg_rng_word((uint32_t)tid, side ? LRG_PURPOSE_RMV : LRG_PURPOSE_ADD, seed, restart, step, k0, k1)) < conf;
            key = lrg_pack_voxel(lrg_voxel_of(__fadd_rn(u.x, c0), prm.resolution), lrg_voxel_of(__fadd_rn(u.y, c1), prm.resolution),
                                 lrg_voxel_of(u.z, prm.resolution));
        }
        sh_key[tid] = (have && take) ? key : LRG_HASH_EMPTY;
        __syncthreads();
        double t = 0.0;
        if (have) {
            bool in = take;
            if (key != LRG_HASH_EMPTY)
                for (int k = 0; k < N; ++k) in |= sh_key[k] == key;     // every lane reads the same word: an LDS broadcast
            const float term = __fdiv_rn(in ? logf(conf) : logf(__fsub_rn(1.f, conf)), (float)prm.n_neighbor);
            t = (double)term;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) sh_part[wave] = t;
        __syncthreads();
        if (tid == 0)
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += sh_part[w];
        __syncthreads();
    }
    if (tid == 0) S->ml_score += total;
}

// ---- (2) dilated voxel-box query with ordered compaction (:221-235) by ONE workgroup ----
// Pass 1 reads the room once, coalesced, 8 points per thread in flight: a byte of flags per (4096-point chunk, thread)
// goes to LDS, a packed (current | candidate << 16) count per (chunk, wavefront) to a table.  One exclusive scan of that
// table (<= 512 entries) gives every (chunk, wavefront) its base; pass 2 takes the flags back from LDS, ranks the lanes of
// a wavefront with a DPP prefix sum and writes the two index lists in index order.
__device__ void lrg_front_query(LrgSlot *S, const LrgRoom *R, const LrgGrowParams &prm, uint8_t *sh_flags, int *sh_tab,
                                int *sh_tabc, int *sh_tabe) {
    __shared__ int wt_c[8], wt_e[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = R->n;
    const int nchunk = (n + LRG_SCAN_CHUNK - 1) / LRG_SCAN_CHUNK;
    const int lo0 = S->mn[0] - 1, lo1 = S->mn[1] - 1, lo2 = S->mn[2] - 1;     // :222-225
    const int hi0 = S->mx[0] + 1, hi1 = S->mx[1] + 1, hi2 = S->mx[2] + 1;
    const uint8_t *cur = S->cur, *visited = R->visited;
    const int32_t *vox = R->voxels;
    for (int c0 = 0; c0 < nchunk; c0 += 2) {            // 8 points per thread per trip (40 registers of loads in flight)
        int cu[8], vi[8], va[8], vb[8], vc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {                   // unconditional loads at clamped indices: no dependent round trips
            const int i = min((c0 + (k >> 2)) * LRG_SCAN_CHUNK + 4 * tid + (k & 3), n - 1);
            cu[k] = cur[i]; vi[k] = visited[i];
            va[k] = vox[3 * i]; vb[k] = vox[3 * i + 1]; vc[k] = vox[3 * i + 2];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = c0 + q;
            if (c < nchunk) {                           // workgroup-uniform
                int fc = 0, fe = 0;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = 4 * q + kk;
                    if (c * LRG_SCAN_CHUNK + 4 * tid + kk < n) {
                        if (cu[k]) fc |= 1 << kk;
                        else if (!vi[k] && va[k] >= lo0 && va[k] <= hi0 && vb[k] >= lo1 && vb[k] <= hi1 && vc[k] >= lo2 && vc[k] <= hi2)
                            fe |= 1 << kk;                                      // :226-228
                    }
                }
                sh_flags[c * LRG_FRONT_THREADS + tid] = (uint8_t)(fc | (fe << 4));
                const int packed = lrg_wave_sum_i32(__popc(fc) | (__popc(fe) << 16));      // <= 256 per half
                if (lane == 0) sh_tab[c * 16 + wave] = packed;
            }
        }
    }
    __syncthreads();
    // exclusive scan of the table in (chunk, wavefront) order by the first 512 threads
    const int nent = nchunk * 16;
    int vcn = 0, ven = 0;
    if (tid < 512 && tid < nent) { const int p = sh_tab[tid]; vcn = p & 0xFFFF; ven = (int)((unsigned)p >> 16); }
    const int ic = lrg_wave_incl_scan_i32(vcn), ie = lrg_wave_incl_scan_i32(ven);
    if (tid < 512 && lane == 63) { wt_c[wave] = ic; wt_e[wave] = ie; }
    __syncthreads();
    int totc = 0, tote = 0;
    {
        int oc = 0, oe = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            if (w < wave) { oc += wt_c[w]; oe += wt_e[w]; }
            totc += wt_c[w]; tote += wt_e[w];
        }
        if (tid < 512) { sh_tabc[tid] = oc + ic - vcn; sh_tabe[tid] = oe + ie - ven; }
    }
    __syncthreads();
    int32_t *cur_idx = S->cur_idx, *cand_idx = S->cand_idx;
    for (int c = 0; c < nchunk; ++c) {
        const int f = sh_flags[c * LRG_FRONT_THREADS + tid];
        const int fc = f & 15, fe = f >> 4;
        const int mine = __popc(fc) | (__popc(fe) << 16);
        const int excl = lrg_wave_incl_scan_i32(mine) - mine;
        int pc = sh_tabc[c * 16 + wave] + (excl & 0xFFFF), pe = sh_tabe[c * 16 + wave] + (int)((unsigned)excl >> 16);
        const int ib = c * LRG_SCAN_CHUNK + 4 * tid;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (fc >> k & 1) cur_idx[pc++] = ib + k;
            if (fe >> k & 1) cand_idx[pe++] = ib + k;
        }
    }
    if (tid == 0) {
        S->pad = 1;
        S->nc = totc;
        S->ne = tote;
        if (tote == 0) {                                                        // :233-235
            S->status = LRG_STOP_NONEIGHBOR; S->last_reason = LRG_STOP_NONEIGHBOR; S->count = totc;
        } else if (prm.max_region_steps > 0 && S->step >= prm.max_region_steps) {
            S->status = LRG_STOP_MAXSTEPS; S->last_reason = LRG_STOP_MAXSTEPS; S->count = totc;
        }
    }
    __syncthreads();
}

// Median of one channel over nc <= 4096 points by one wavefront, 64 keys per lane; the gather goes in two batches of 32
// so that index and key registers of a batch are not all live at once (1024-thread workgroups have 128 VGPRs per lane).
__device__ __forceinline__ float lrg_median_wave_r64(const float *pts, const int32_t *idx, int F, int nc) {
    const int lane = lrg_lane();
    uint32_t key[64];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int id[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) id[r] = idx[min((h * 32 + r) * 64 + lane, nc - 1)];
#pragma unroll
        for (int r = 0; r < 32; ++r) key[h * 32 + r] = ((h * 32 + r) * 64 + lane < nc) ? lrg_f2key(pts[(long)id[r] * F]) : 0xFFFFFFFFu;
    }
    return lrg_select_regs<64>(key, nc);
}

// ---- (3) medians, subset sampling, gather of the distinct rows into the packed arrays ----
__device__ void lrg_front_prepare(const LrgSlot *S, const LrgRoom *R, int s, const LrgGrowParams &prm, const LrgFrontArgs &a,
                                  int (*sh_src)[LRG_FRONT_MAXSAMPLE], float *sh_c, int *sh_med) {
    __shared__ int sh_off[2];
    const int tid = threadIdx.x, bd = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int F = prm.feature_size, Ni = prm.n_inlier, Nn = prm.n_neighbor;
    const int nc = S->nc, ne = S->ne;
    const int rin = min(nc, Ni), rnb = min(ne, Nn);
    const float *points = R->points;
    const int32_t *obj = R->obj_id;
    if (tid == 0) {
        const int oi = atomicAdd(&a.counters[0], rin), on = atomicAdd(&a.counters[1], rnb);
        sh_off[0] = oi; sh_off[1] = on;
        a.slot_rows[4 * s + 0] = rin; a.slot_rows[4 * s + 1] = rnb; a.slot_rows[4 * s + 2] = oi; a.slot_rows[4 * s + 3] = on;
    }
    // ---- subset sampling (:237-240, :249-252): positions -> source indices (independent of the centre) ----
    const uint32_t k0 = prm.rng_seed, k1 = (uint32_t)R->room_id;
    for (int u = tid; u < Ni + Nn; u += bd) {
        const int side = u >= Ni, j = side ? u - Ni : u;
        const int n = side ? ne : nc, k = side ? Nn : Ni;
        const int pos = (int)lrg_sample_position((uint32_t)j, (uint32_t)n, (uint32_t)k, side ? LRG_PURPOSE_NEIGHBOR : LRG_PURPOSE_INLIER,
                                                 (uint32_t)S->seed, (uint32_t)S->restart, (uint32_t)S->step, k0, k1);
        (side ? a.sample_nb : a.sample_in)[(long)s * k + j] = pos;
        if (j < min(n, k)) sh_src[side][j] = (side ? S->cand_idx : S->cur_idx)[pos];     // rows past min(n, k) are copies of these
    }
    if (tid < 16) sh_c[tid] = 0.f;
    __syncthreads();
    TRACE2(s, 4);
    // ---- centre (:241): per-channel median of ALL current points, channels 0,1,6.. (:243-247) ----
    if (nc <= 4096) {
        // one wavefront per centred channel, keys in registers, no barrier
        if (wave < 9) {
            const int ch = lrg_centred_channel(wave, F);
            if (ch >= 0) {
                const LrgChanSrc cs = lrg_chan_src(R, wave, ch, F);
                const float m = nc <= 256 ? lrg_median_wave_r<4>(cs.base, S->cur_idx, cs.stride, nc)
                              : nc <= 1024 ? lrg_median_wave_r<16>(cs.base, S->cur_idx, cs.stride, nc)
                                           : lrg_median_wave_r64(cs.base, S->cur_idx, cs.stride, nc);
                if (lane == 0) sh_c[ch] = m;
            }
        }
    } else {
        for (int y = 0; y < 9; ++y) {
            const int ch = lrg_centred_channel(y, F);
            if (ch < 0) continue;
            __syncthreads();
            if (tid < 64) sh_med[tid] = tid == 0 ? -1 : 0;
            __syncthreads();
            const LrgChanSrc cs = lrg_chan_src(R, y, ch, F);
            float m;
            if (nc <= 16 * 1024) m = lrg_median_block_regs<16>(cs.base, S->cur_idx, cs.stride, nc, sh_med);
            else if (nc <= LRG_MED_REGS) m = lrg_median_block_regs<48>(cs.base, S->cur_idx, cs.stride, nc, sh_med);
            else {
                const int k2 = nc >> 1, k1r = (nc & 1) ? k2 : k2 - 1;
                uint32_t ka, kb;
                lrg_select2(nullptr, false, cs.base, S->cur_idx, cs.stride, 0, nc, k1r, k2, sh_med, &ka, &kb);
                const float lo = lrg_key2f(ka), hi = lrg_key2f(kb);
                m = (nc & 1) ? hi : __fmul_rn(__fadd_rn(lo, hi), 0.5f);
            }
            if (tid == 0) sh_c[ch] = m;
        }
    }
    __syncthreads();
    TRACE2(s, 5);
    if (tid < 16) a.center[s * 16 + tid] = sh_c[tid];       // the next update un-centres x, y with it (:271,:275)
    const int offi = sh_off[0], offn = sh_off[1];
    // ---- per-row tags: owning slot and the ground-truth flags input_remove / input_add (:230-231,:248,:254) ----
    const int target = S->target;
    // (and the slot's own copy of what the next update needs of each row: its first three columns as stored -- centred -- and the flag)
    // (columns 0..2 follow from the gather loop below)
    for (int j = tid; j < rin; j += bd) {
        a.row_slot_in[offi + j] = s;
        reinterpret_cast<float *>(a.upd_in)[((long)s * Ni + j) * 4 + 3] = (obj && obj[sh_src[0][j]] != target) ? 1.f : 0.f;
    }
    for (int j = tid; j < rnb; j += bd) {
        a.row_slot_nb[offn + j] = s;
        reinterpret_cast<float *>(a.upd_nb)[((long)s * Nn + j) * 4 + 3] = (obj && obj[sh_src[1][j]] == target) ? 1.f : 0.f;
    }
    TRACE2(s, 6);
    // ---- gather + centre (:242-254), element-wise so that loads and stores of a row are contiguous across lanes ----
    for (int side = 0; side < 2; ++side) {
        const int k = side ? rnb : rin;
        float *out = side ? a.x_nb + (long)offn * F : a.x_in + (long)offi * F;
        float *upd = reinterpret_cast<float *>(side ? a.upd_nb : a.upd_in) + (long)s * (side ? Nn : Ni) * 4;
        const int nel = k * F;
        for (int e0 = tid; e0 < nel; e0 += 8 * bd) {       // 8 independent row loads in flight per thread
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = min(e0 + u * bd, nel - 1);
                const int j = e / F, f = e - j * F;
                v[u] = __fsub_rn(points[(long)sh_src[side][j] * F + f], sh_c[f]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * bd;
                if (e < nel) {
                    out[e] = v[u];
                    const int j = e / F, f = e - j * F;
                    if (f < 3) upd[4 * j + f] = v[u];      // the slot's own copy of columns 0..2 for the next update
                }
            }
        }
    }
}

// MODE bits: 1 = mask update + stop decision, 2 = commit / next seed for single-slot groups (greedy growing), 4 = box query +
// preparation.  Greedy growing runs all three in one launch; restart groups put lrg_advance_kernel (one workgroup per
// GROUP) between a MODE 1 and a MODE 4 launch.
template <int MODE>
__global__ __launch_bounds__(LRG_FRONT_THREADS) void lrg_front_kernel(LrgSlot *slots, LrgRoom *rooms, int n_slots,
                                                                       LrgGrowParams prm, LrgFrontArgs a) {
    __shared__ uint8_t sh_flags[LRG_FRONT_MAXCHUNK * LRG_FRONT_THREADS];
    __shared__ int sh_tab[512], sh_tabc[512], sh_tabe[512];
    __shared__ int sh_src[2][LRG_FRONT_MAXSAMPLE];
    __shared__ float sh_c[16];
    __shared__ int red[16 * 8];
    const int s = blockIdx.x, tid = threadIdx.x;
    LrgSlot *S = &slots[s];
    const int room = S->room;
    if ((MODE & 1) && a.pooled)      // the last evaluation's pooled feature has been consumed: zero for the next one
        for (int c = tid; c < a.pooled_stride; c += blockDim.x) a.pooled[(long)s * a.pooled_stride + c] = 0.f;
    if (room < 0) {
        if ((MODE & 4) && tid == 0) { a.slot_rows[4 * s + 0] = 0; a.slot_rows[4 * s + 1] = 0; }
        return;
    }
    LrgRoom *R = &rooms[room];
    TRACE2(s, 0);
    if constexpr ((MODE & 1) != 0) {
        if (prm.scoring == 1 && S->status == LRG_ACTIVE) lrg_front_ml_score(S, R, s, prm, a);      // (also restarts with one slot per group, MODE 7)
    }
    if ((MODE & 1) && S->status == LRG_ACTIVE) lrg_front_update(S, R, s, prm, a, &sh_src[0][0], red);
    TRACE2(s, 1);
    if (MODE & 2) {
        __syncthreads();
        lrg_advance_group<false>(slots, rooms, n_slots, prm, a.stats, s);
        __syncthreads();
    }
    TRACE2(s, 2);
    if (MODE & 4) {
        bool active = S->status == LRG_ACTIVE;
        if (active && S->pad != 1) {
            lrg_front_query(S, R, prm, sh_flags, sh_tab, sh_tabc, sh_tabe);
            active = S->status == LRG_ACTIVE;
        }
        TRACE2(s, 3);
        if (!active) {
            if (tid == 0) { a.slot_rows[4 * s + 0] = 0; a.slot_rows[4 * s + 1] = 0; }
            return;
        }
        lrg_front_prepare(S, R, s, prm, a, sh_src, sh_c, red);
        TRACE2(s, 7);
#if LRG_TRACE
        if (tid == 0 && g_lrg_trace2) { g_lrg_trace2[(long)s * 16 + 8] = S->nc; }
#endif
    }
}

// =================================================================================================
// Greedy growing (group_size = restarts = 1), n_inlier / n_neighbor <= 512, rooms with packed voxel words: the whole
// front of an iteration in one launch, written for the number of DEPENDENT memory round trips (a first-touch load costs
// ~1 us here: everything a kernel reads was written by a kernel on another XCD), not for the amount of work:
//   * everything addressed by the slot number alone is requested at once (slot, row table, centre, this thread's sample);
//   * the 512 add slots and the 512 remove slots are decided side by side by the two halves of the workgroup, and the
//     old index list with its voxel words is fetched while their hash probes are in flight;
//   * commit and reset walk the region's index lists (O(region)) instead of the room;
//   * the box query reads 4 points per load (mask word, visited word, four packed voxel words);
//   * regions above LRG_FRONT_SMALL points leave their nine channel medians to lrg_front_big_kernel, one workgroup per
//     (slot, channel) -- the single slow step whose cost grows with the region (its last-arriving workgroup gathers).
// =================================================================================================
#ifndef LRG_FRONT_SMALL
#define LRG_FRONT_SMALL 0
#endif
                                // regions above this many points get their medians from lrg_front_big_kernel.  0 = all of them:
                                // that launch runs in (almost) every iteration anyway -- with 68 slots some region is nearly
                                // always large -- and its duration is set by the largest region, so the small ones ride along
                                // for free and the front kernel loses its median phase (trace: 8.7 k of 44 k cycles per slot)

#ifndef LRG_QUERY_TRIP
#define LRG_QUERY_TRIP 4
#endif
#define LRG_PVX(p) ((int)((p) & 0x7FFu))
#define LRG_PVY(p) ((int)(((p) >> 11) & 0x7FFu))
#define LRG_PVZ(p) ((int)((p) >> 22))

// tags + gather of the slot's distinct rows (shared by the front kernel and the last workgroup of lrg_front_big_kernel)
// (first, nthreads): the threads [first, first + nthreads) of the workgroup do the work, so that other wavefronts can compute
// the medians meanwhile.  The rows are stored UNCENTRED: the branch kernels subtract the centre while staging them
// (lrg_forward_packed, `center`), the next update re-derives the centred value -- so the gather does not wait for the medians.
// v / d for 0 <= v < 2^23, d >= 1 with rcp = 1.0f / d: float estimate, one correction step either way
__device__ __forceinline__ int lrg_div_small(int v, int d, float rcp) {
    int q = (int)(((float)v + 0.5f) * rcp);
    if (q * d > v) --q;
    else if ((q + 1) * d <= v) ++q;
    return q;
}
// A slot's packed rows are allocated in multiples of LRG_ROW_PAD rows, the padding filled with copies of the set's last row (the
// copies the reference itself pads a set with, :240,:252: the max-pool ignores duplicates, nobody reads their logits).  A 32-row
// tile of the network then holds at most 32 / LRG_ROW_PAD runs of rows of different slots: every run costs its epilogue passes
// a masked maximum and a bias of its own (a tile of 8 runs lived 50 k cycles against 27 k for one run, profiles/r02_head_pass_stamps.txt),
// and the slowest tile is what a launch lasts.
#ifndef LRG_ROW_PAD
#define LRG_ROW_PAD 8
#endif
#define LRG_PAD_ROWS_TO(r, p) (((r) + (p) - 1) / (p) * (p))
#define LRG_PAD_ROWS(r) LRG_PAD_ROWS_TO(r, LRG_ROW_PAD)

// Workgroup barrier for hand-overs through LDS only: waits for this wavefront's LDS operations, not for the acknowledgement of
// its global stores (__syncthreads() does, ~1.5 k cycles after a burst of stores that nobody in the workgroup reads back).
#define LRG_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Both sides in one element space (inlier rows, then neighbour rows): one round trip for all rows, and the obj_id words of the
// ground-truth flags input_remove / input_add (:230-231,:248,:254) ride along (element 0 of each row).  Besides the packed rows,
// every row leaves x, y, z (as stored: uncentred) and its flag in the slot's own upd_* arrays for the next mask update.
// U = loads in flight per thread: two full sets (2 x 512 rows x 13 floats = 13 k elements) still go in one trip with 13.
template <int U, int PAD, bool COH>
__device__ __forceinline__ void lrg_front_gather_rows(int target, const float *points, const int32_t *obj, int s, int F, int ni, int nn,
                                                      const LrgFrontArgs &a, const int (*sh_src)[512], int rin, int rnb, int offi, int offn,
                                                      int tid, int bd) {
    const int rin_p = LRG_PAD_ROWS_TO(rin, PAD), rnb_p = LRG_PAD_ROWS_TO(rnb, PAD);      // (rows past the set: copies of its last row)
    const int nel_in = rin_p * F, nel = nel_in + rnb_p * F;
    const float rF = 1.0f / (float)F;
    float *out_in = a.x_in + (long)offi * F, *out_nb = a.x_nb + (long)offn * F;
    float *upd_in = reinterpret_cast<float *>(a.upd_in) + (long)s * ni * 4, *upd_nb = reinterpret_cast<float *>(a.upd_nb) + (long)s * nn * 4;
    for (int e0 = tid; e0 < nel; e0 += U * bd) {
        float v[U];
        int ob[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u * bd, nel - 1);
            const int side = e >= nel_in ? 1 : 0, l = e - (side ? nel_in : 0);
            const int j = lrg_div_small(l, F, rF), f = l - j * F;
            const int src = sh_src[side][min(j, (side ? rnb : rin) - 1)];
            v[u] = points[(long)src * F + f];
            ob[u] = (f == 0 && obj) ? obj[src] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * bd;                     // (row and column worked out again: registers are for the loads in flight)
            if (e < nel) {
                const int side = e >= nel_in ? 1 : 0, l = e - (side ? nel_in : 0);
                const int j = lrg_div_small(l, F, rF), f = l - j * F;
                if constexpr (COH) lrg_st_coh((side ? out_nb : out_in) + l, v[u]); else (side ? out_nb : out_in)[l] = v[u];
                if (j < (side ? rnb : rin)) {
                    float *upd = (side ? upd_nb : upd_in) + 4 * j;
                    if (f < 3) upd[f] = v[u];
                    if (f == 0) upd[3] = (obj && (side ? ob[u] == target : ob[u] != target)) ? 1.f : 0.f;
                }
            }
        }
    }
}

// The same gather for the free-running kernel with rows at a 64-byte stride (LrgFrontArgs.rows16): an item = (row, quarter of the row) -- up to four
// source floats in, ONE 16-byte write-through store out.  As 13 dword stores per row every element was a fabric write of its own (MI355X_MICROARCH.md:
// a dword `sc1` store costs ~6 x the time per byte of a 16-byte one): the 1 024-row gather of a 100 k-point scene took 10.6 us of a 38 us front step, the
// ~220 rows of an Area-5 step 3.3 of 23.6 (profiles/r04_kitti_breakdown.log); the tile teams stage such a row with four 16-byte loads.
template <int U, int PAD>
__device__ __forceinline__ void lrg_front_gather_rows16(int target, const float *points, const int32_t *obj, int s, int F, int ni, int nn,
                                                        const LrgFrontArgs &a, const int (*sh_src)[512], int rin, int rnb, int offi, int offn,
                                                        int tid, int bd, int tbi = -1, int tbn = -1) {
    // (tbi / tbn >= 0: the side's rows beyond its last full tile go to the shared rows from there, unpadded; else the side is padded to whole tiles in its own place)
    const int rin_p = tbi >= 0 ? rin : LRG_PAD_ROWS_TO(rin, PAD), rnb_p = tbn >= 0 ? rnb : LRG_PAD_ROWS_TO(rnb, PAD);      // (rows past the set: copies of its last row)
    const int nit_in = rin_p * 4, nit = nit_in + rnb_p * 4;
    const int fi = tbi >= 0 ? (rin & ~31) : INT_MAX, fn = tbn >= 0 ? (rnb & ~31) : INT_MAX;      // first row of the tail
    float *out_in = a.x_in + (long)offi * 16, *out_nb = a.x_nb + (long)offn * 16;
    float4 *upd_in = a.upd_in + (long)s * ni, *upd_nb = a.upd_nb + (long)s * nn;
    for (int i0 = tid; i0 < nit; i0 += U * bd) {
        float v[U][4];
        int ob[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = min(i0 + u * bd, nit - 1);
            const int side = it >= nit_in ? 1 : 0, l = it - (side ? nit_in : 0);
            const int j = l >> 2, q = l & 3;
            const int src = sh_src[side][min(j, (side ? rnb : rin) - 1)];
            const float *p = points + (long)src * F + 4 * q;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = 4 * q + k < F ? p[k] : 0.f;
            ob[u] = (q == 0 && obj) ? obj[src] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = i0 + u * bd;
            if (it < nit) {
                const int side = it >= nit_in ? 1 : 0, l = it - (side ? nit_in : 0);
                const int j = l >> 2, q = l & 3;
#ifndef LRG_EXP_NO_GATHER_STORE      // (experiment switch, --policy gt only: what the rows' write-through stores cost the front step)
                if (j >= (side ? fn : fi)) {      // a tail row: to the shared rows, with its tag (the shared tile finds its runs of rows by them)
                    const long row = (long)a.tail_row0 + (side ? tbn : tbi) + (j - (side ? fn : fi));
                    lrg_st_coh4((side ? a.x_nb : a.x_in) + row * 16, (unsigned)q * 16u, make_float4(v[u][0], v[u][1], v[u][2], v[u][3]));
                    if (q == 0) lrg_st_coh(&(side ? a.row_slot_nb : a.row_slot_in)[row], s);
                } else
                lrg_st_coh4(side ? out_nb : out_in, (unsigned)l * 16u, make_float4(v[u][0], v[u][1], v[u][2], v[u][3]));
#endif
                if (q == 0 && j < (side ? rnb : rin))
                    (side ? upd_nb : upd_in)[j] = make_float4(v[u][0], v[u][1], v[u][2], (obj && (side ? ob[u] == target : ob[u] != target)) ? 1.f : 0.f);
            }
        }
    }
}

template <int PAD, bool COH>
__device__ __forceinline__ void lrg_front_gather(int target, const float *points, const int32_t *obj, int s, int F, int ni, int nn,
                                                 const LrgFrontArgs &a, const int (*sh_src)[512], int rin,
                                                 int rnb, int offi, int offn, int first, int nthreads, int tbi = -1, int tbn = -1) {
    const int tid = (int)threadIdx.x - first, bd = nthreads;
    if (tid < 0 || tid >= nthreads) return;
    if constexpr (!COH) {         // (the free-running kernel: a slot's rows have a fixed place, the tags were written once)
        for (int j = tid; j < LRG_PAD_ROWS_TO(rin, PAD); j += bd) a.row_slot_in[offi + j] = s;
        for (int j = tid; j < LRG_PAD_ROWS_TO(rnb, PAD); j += bd) a.row_slot_nb[offn + j] = s;
    }
    if constexpr (COH) {
        if (a.rows16) {
            const int nit = (LRG_PAD_ROWS_TO(rin, PAD) + LRG_PAD_ROWS_TO(rnb, PAD)) * 4;
            if (nit <= bd) lrg_front_gather_rows16<1, PAD>(target, points, obj, s, F, ni, nn, a, sh_src, rin, rnb, offi, offn, tid, bd, tbi, tbn);
            else if (nit <= 2 * bd) lrg_front_gather_rows16<2, PAD>(target, points, obj, s, F, ni, nn, a, sh_src, rin, rnb, offi, offn, tid, bd, tbi, tbn);
            else lrg_front_gather_rows16<4, PAD>(target, points, obj, s, F, ni, nn, a, sh_src, rin, rnb, offi, offn, tid, bd, tbi, tbn);
            return;
        }
    }
    const int nel = (LRG_PAD_ROWS_TO(rin, PAD) + LRG_PAD_ROWS_TO(rnb, PAD)) * F;
    if (nel <= 2 * bd) lrg_front_gather_rows<2, PAD, COH>(target, points, obj, s, F, ni, nn, a, sh_src, rin, rnb, offi, offn, tid, bd);
    else if (nel <= 6 * bd) lrg_front_gather_rows<6, PAD, COH>(target, points, obj, s, F, ni, nn, a, sh_src, rin, rnb, offi, offn, tid, bd);
    else lrg_front_gather_rows<13, PAD, COH>(target, points, obj, s, F, ni, nn, a, sh_src, rin, rnb, offi, offn, tid, bd);   // 2 x 512 rows x 13 floats
}

// voxel -> point index: one load from the room's dense grid when it has one, else the probe chain of the hash table
struct LrgVoxIndex {
    const int32_t *grid;
    int gx, gy, gz, ox, oy, oz;
    const uint64_t *hkeys;
    const int32_t *hvals;
    int hmask;
};
__device__ __forceinline__ int lrg_voxel_index(const LrgVoxIndex &I, int vx, int vy, int vz) {
    if (I.grid) {
        const int x = vx - I.ox, y = vy - I.oy, z = vz - I.oz;
        if ((unsigned)x >= (unsigned)I.gx || (unsigned)y >= (unsigned)I.gy || (unsigned)z >= (unsigned)I.gz) return -1;
        return I.grid[((long)z * I.gy + y) * I.gx + x];
    }
    return lrg_hash_lookup(I.hkeys, I.hvals, I.hmask, lrg_pack_voxel(vx, vy, vz));
}
#ifndef LRG_GRID_QUERY_CELLS
#define LRG_GRID_QUERY_CELLS (16 * LRG_FRONT_THREADS)   // dilated boxes up to this many voxels are answered from the grid (measured: a
                                                        // limit of 32 Ki with 12 or 16 cells per trip gains nothing over the room-wide pass)
#endif

// All centred channels of a region by the slot's own 1024-thread workgroup, after its gather (sh: LRG_RADIX_LDS_INTS(9) ints):
// the medians used to be a launch of their own between the front kernel and the branch stacks -- ~9 us of a chain of dependent,
// latency-bound launches, 117.5 -> 107.5 us per iteration without it (tools/r02_nobig.sh) -- whose duration was the largest region's.
// Up to 4 Ki points the nine channels go through ONE radix select together (one index load, the keys of all channels in flight at
// once, shared barriers); up to 16 Ki in three groups of three; above, channel after channel by bisection over the list in memory
// (slow, exact, and a region of that size in a room of at most 131 072 points is a rarity: rooms above 65 536 points run the
// chunk-parallel lrg_grow_step by default).
__device__ __noinline__ void lrg_front_all_medians(const LrgRoom *R, const int32_t *cur_idx, int nc, int F, int *sh, float *sh_c) {
    const int tid = threadIdx.x;
    const float *cm = R->chan_major;
    const float *base = cm ? cm : R->points;
    const int stride = cm ? 1 : F;
    if (nc <= 4096) {
        int chs[9];
        float m[9];
#pragma unroll
        for (int y = 0; y < 9; ++y) { const int ch = lrg_centred_channel(y, F); chs[y] = ch < 0 ? -1 : cm ? y * R->chan_stride : ch; }
        lrg_median_block_radix<4, LRG_FRONT_THREADS, 9>(base, chs, cur_idx, stride, nc, sh, m);
        if (tid == 0) {
#pragma unroll
            for (int y = 0; y < 9; ++y) { const int ch = lrg_centred_channel(y, F); if (ch >= 0) sh_c[ch] = m[y]; }
        }
    } else if (nc <= 16 * 1024) {
        for (int g = 0; g < 3; ++g) {
            int chs[3];
            float m[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { const int y = 3 * g + c, ch = lrg_centred_channel(y, F); chs[c] = ch < 0 ? -1 : cm ? y * R->chan_stride : ch; }
            __syncthreads();
            lrg_median_block_radix<16, LRG_FRONT_THREADS, 3>(base, chs, cur_idx, stride, nc, sh, m);
            if (tid == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { const int ch = lrg_centred_channel(3 * g + c, F); if (ch >= 0) sh_c[ch] = m[c]; }
            }
        }
    } else {
        for (int y = 0; y < 9; ++y) {
            const int ch = lrg_centred_channel(y, F);
            if (ch < 0) continue;
            __syncthreads();
            if (tid < 64) sh[tid] = tid == 0 ? -1 : 0;
            __syncthreads();
            const LrgChanSrc cs = lrg_chan_src(R, y, ch, F);
            const int k2 = nc >> 1, k1r = (nc & 1) ? k2 : k2 - 1;
            uint32_t ka, kb;
            lrg_select2(nullptr, false, cs.base, cur_idx, cs.stride, 0, nc, k1r, k2, sh, &ka, &kb);
            const float lo = lrg_key2f(ka), hi = lrg_key2f(kb);
            if (tid == 0) sh_c[ch] = (nc & 1) ? hi : __fmul_rn(__fadd_rn(lo, hi), 0.5f);
        }
    }
}

// The median of centred channel y of slot S's current points by one 1024-thread workgroup (the body of lrg_front_big_kernel; CAP =
// bracket buffer of the sampled selection, so that `sh` fits where the caller has room).  Valid in thread 0.
template <int CAP>
__device__ __forceinline__ float lrg_big_median(const int32_t *cur_idx, const LrgRoom *R, int nc, int y, int ch, int F, int *sh) {
    const int tid = threadIdx.x;
    const LrgChanSrc cs = lrg_chan_src(R, y, ch, F);
    if (nc <= 1024) {                                // one wavefront, keys in registers, no barrier
        float mw = 0.f;
        if (tid < 64) mw = nc <= 256 ? lrg_median_wave_r<4>(cs.base, cur_idx, cs.stride, nc) : lrg_median_wave_r<16>(cs.base, cur_idx, cs.stride, nc);
        return mw;
    }
    float m;
    const int chs[1] = {0};
    float mm[1];
    if (nc <= 4096) { lrg_median_block_radix<4, 1024, 1>(cs.base, chs, cur_idx, cs.stride, nc, sh, mm); m = mm[0]; }
    else if (nc <= 16 * 1024) { lrg_median_block_radix<16, 1024, 1>(cs.base, chs, cur_idx, cs.stride, nc, sh, mm); m = mm[0]; }
    else if (nc <= LRG_MED_REGS) m = lrg_median_block_sampled<48, 1024, CAP>(cs.base, cur_idx, cs.stride, nc, sh);
    else {
        if (tid < 64) sh[tid] = tid == 0 ? -1 : 0;
        __syncthreads();
        const int k2 = nc >> 1, k1r = (nc & 1) ? k2 : k2 - 1;
        uint32_t ka, kb;
        lrg_select2(nullptr, false, cs.base, cur_idx, cs.stride, 0, nc, k1r, k2, sh, &ka, &kb);
        const float lo = lrg_key2f(ka), hi = lrg_key2f(kb);
        m = (nc & 1) ? hi : __fmul_rn(__fadd_rn(lo, hi), 0.5f);
    }
    return m;
}

// The greedy front kernel is accounted 128 VGPRs (it needs 88): four of its waves then fill a SIMD's register file, so a slot's
// workgroup has its CU to itself.  With two lanes the other lane's tile workgroups otherwise move in beside it (160 VGPRs are left
// per SIMD), slow it down, and a launch lasts as long as its slowest slot: 566.6 k -> 586.2 k instance-steps/s (115.0 -> 111.2 us per
// iteration; no change on one lane; tools/r02_excl.sh).
#ifndef LRG_FRONT_EXCLUSIVE
#define LRG_FRONT_EXCLUSIVE 1
#endif
#ifndef LRG_BIG_EXCLUSIVE
#define LRG_BIG_EXCLUSIVE 0
#endif
// LDS of the greedy front (a struct, so that the free-running kernel can lay it over the same bytes its tile teams use)
#define LRG_FRONT_FLAG_BYTES (LRG_FRONT_MAXCHUNK * LRG_FRONT_THREADS > 4 * LRG_RADIX_LDS_INTS(9) ? LRG_FRONT_MAXCHUNK * LRG_FRONT_THREADS : 4 * LRG_RADIX_LDS_INTS(9))
struct LrgFrontShared {
    // flags of the room-wide pass / index bitmaps of the grid query / histograms of the nine-channel radix select
    __attribute__((aligned(16))) uint8_t flags[LRG_FRONT_FLAG_BYTES];
    int tab[512], tabc[512], tabe[512];
    int src[2][512];         // during the update: [0] = indices switched on by this step
    __attribute__((aligned(16))) float c[16];
    int red[16 * 8];
    int i[8];                // 0 updated, 1 added count, 2 status, 3 seed-search minimum, 4 probe count, 5 add_acc, 6 remove_acc
    int list[32];
    int wt_c[16], wt_e[16];  // (the grid query scans over all 16 wavefronts)
    int box[8];              // bounding box of the mask (S->mn, S->mx) for the box query, [6] count, [7] target
    int off[2];
    int tail[4];             // shared tail tiles: [0] / [1] first shared row of the inlier / neighbour tail (-1: none), [2] / [3] rows of a failed reservation that lie
                             // inside the shared rows (dead rows of the last shared tile, for the serving loop to account)
    int upd[4];              // the mask update: [0] != 0 = some sample slot's re-derived voxel is not its own point's (the update then takes the general form).  Zero
                             // between two updates: cleared by whoever declares the struct, and again by every update after it has been read
};

// ---- speculation (LrgFrontArgs.spec_k = K > 1): several regions of ONE room in flight ----
// The reference grows a room's regions one after the other: the seeds are visited in curvature order, a visited point is no seed and no
// candidate (test_region_grow.py:186-188,:227-228), and a region marks its points visited when it stops (:210-217).  A room is a chain of
// dependent steps, and a GPU that holds one room (BASELINE configs 3 and 5: one room or scene per GPU) runs one such chain.  Here the K slots of
// a group, all served by ONE front workgroup, grow the regions of the room's next K unvisited seeds side by side:
//   * seeds are handed out in order (R->seed_cursor), so the positions in flight are consecutive unvisited positions of the order;
//   * regions are COMMITTED in seed order: a region that has stopped waits (LRG_PENDING, its members kept in its index list) until no
//     slot of the group holds an earlier position;
//   * a commit voids every later region in flight that could have seen one of the committed points as a candidate: every box query of a
//     region lies inside [seq_mn - 1, seq_mx + 1] (its running bounding box, dilated: :222-225,:302-303), so a committed point inside that
//     box of slot t sets t's void flag; t drops its region (at its next turn: an evaluation in flight cannot be recalled) and grows again
//     from the same seed if that is still unvisited, else from the next seed handed out.  A region that was not voided saw, in every
//     query, exactly the visited flags the sequential loop would have shown it: committed points outside its boxes never were candidates, and
//     its own members cannot have been committed by another region (a member was a candidate first: inside a box).
// The random stream is keyed by (room, seed point, step), not by slot or order of execution, so the regions, their order in the log and the
// cluster ids (assigned at commit) are those of the sequential loop: identical labels (tests/test_gpu_speculation.py).
#ifdef LRG_SPEC_TRACE      // (debug build: events of the speculation protocol into LrgAsyncBuffers.work + 8: [0] count, then 4 words per event -- tools/spec_debug.py allocates 4 x 65536)
#define LRG_SPEC_EV(a, type, slot, v0, v1, v2) do { if ((a).spec_stats) { const unsigned long long e_ = atomicAdd(&(a).spec_stats[4], 1ULL); if (e_ < 65536) { \
    unsigned long long *p_ = (a).spec_stats + 5 + 4 * e_; p_[0] = ((unsigned long long)(type) << 32) | (unsigned)(slot); p_[1] = (unsigned long long)(long long)(v0); \
    p_[2] = (unsigned long long)(long long)(v1); p_[3] = (unsigned long long)(long long)(v2); } } } while (0)
#else
#define LRG_SPEC_EV(a, type, slot, v0, v1, v2) do {} while (0)
#endif
// Every decision of the protocol that all wavefronts must take alike is taken by THREAD 0 from loads of its own (past the L1) and handed to the others through LDS:
// a "uniform" global load is not uniform in time -- wavefronts that read a word while another thread's store to it is on its way see different values, take
// different branches and meet at different barriers (seen: the inlier half's sampling one phase behind the rest).
__device__ __forceinline__ bool lrg_spec_is_head(LrgFrontShared &SH, const LrgSlot *slots, int g0, int K, int s, int my_pos) {
    __syncthreads();                             // (SH.i[7] may still be read from the last broadcast)
    if (threadIdx.x == 0) {
        bool head = true;
        for (int t = g0; t < g0 + K; ++t)
            if (t != s && lrg_ld_coh(&slots[t].spec_pos) < my_pos) head = false;
        SH.i[7] = head ? 1 : 0;
    }
    __syncthreads();
    return SH.i[7] != 0;
}

// Commit of the region slot s holds in its index list (entries that are no longer members -- removed by the last update -- are skipped):
// :210-217, the log entry, and the void flags of the later regions in flight.  All threads; ends with a barrier.
__device__ __noinline__ void lrg_spec_commit(LrgFrontShared &SH, LrgSlot *slots, LrgRoom *R, int g0, int K, int s, const LrgGrowParams &prm, const LrgFrontArgs &a,
                                             int nlist, int count) {
    const int tid = threadIdx.x;
    LrgSlot *S = &slots[s];
    int *sh_box = SH.tabc;                    // [K - 1 <= 15][6] dilated running boxes of the later regions in flight, [96 + b] their slots (LDS free between two phases of the front)
    int *sh_hit = SH.i + 7;                   // bit t: slot g0 + t is void
    const int labeled = count > prm.cluster_threshold;                                       // :213   (nlist, count: in registers from the caller)
    const int cid = R->next_cluster_id;
    uint8_t *cur = S->cur, *visited = R->visited;
    int32_t *label = R->label;
    const int32_t *cur_idx = S->cur_idx;
    const int32_t *vox = R->voxels;
    // the dilated running boxes of the other regions in flight, through LDS (thread 0 reads them past the L1)
    if (tid == 0) {
        int k = 0;
        for (int t = g0; t < g0 + K; ++t) {
            if (t == s || lrg_ld_coh(&slots[t].spec_pos) == INT_MAX) continue;
            const LrgSlot *T = &slots[t];
            for (int d = 0; d < 3; ++d) { sh_box[6 * k + d] = lrg_ld_coh(&T->seq_mn[d]) - 1; sh_box[6 * k + 3 + d] = lrg_ld_coh(&T->seq_mx[d]) + 1; }
            sh_box[96 + k] = t - g0;
            ++k;
        }
        sh_box[120] = k;
        *sh_hit = 0;
    }
    __syncthreads();
    const int nb = sh_box[120];
    int hit = 0;
    for (int i = tid; i < nlist; i += LRG_FRONT_THREADS) {
        const int id = lrg_ld_coh(&cur_idx[i]);                                              // (the list may have been written a moment ago, by other wavefronts)
        if (!cur[id]) continue;
        visited[id] = 1; if (labeled) label[id] = cid; cur[id] = 0;                          // :212,:214 + reset
        if (nb) {
            const int x = vox[3 * (long)id], y = vox[3 * (long)id + 1], z = vox[3 * (long)id + 2];
            for (int b = 0; b < nb; ++b)
                if (x >= sh_box[6 * b] && x <= sh_box[6 * b + 3] && y >= sh_box[6 * b + 1] && y <= sh_box[6 * b + 4] && z >= sh_box[6 * b + 2] && z <= sh_box[6 * b + 5])
                    hit |= 1 << sh_box[96 + b];
        }
    }
    if (hit) atomicOr(sh_hit, hit);
    // (__syncthreads() does not wait for global stores -- `s_waitcnt lgkmcnt(0); s_barrier` -- and the other wavefronts read these flags right behind the barrier:
    //  the seed search that follows must see this region's points as visited)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
#ifdef LRG_SPEC_VOID_ALL      // (debug switch: every commit voids every other region in flight -- speculation without any benefit, and without the box test)
        int h = 0;
        for (int t = 0; t < K; ++t) if (g0 + t != s && slots[g0 + t].spec_pos != INT_MAX) h |= 1 << t;
#else
        const int h = *sh_hit;
#endif
        for (int t = 0; t < K; ++t)
            if (h >> t & 1) slots[g0 + t].spec_flags |= 1;
        int32_t *log = R->region_log + LRG_LOG_WORDS * (long)R->n_regions;
        log[0] = S->seed; log[1] = S->steps_total; log[2] = count; log[3] = S->last_reason; log[4] = labeled; log[5] = 0;
        log[6] = S->acc_add; log[7] = S->acc_rmv;
        R->n_regions += 1;
        if (labeled) R->next_cluster_id = cid + 1;                                           // :215
        if (a.stats) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&a.stats[0]), 1ULL);
        }
        if (h && a.spec_stats) atomicAdd(&a.spec_stats[0], (unsigned long long)__popc(h));
        LRG_SPEC_EV(a, 3, s, S->seed, (count << 8) | S->last_reason, h);
        S->spec_pos = INT_MAX;
        S->status = LRG_WAIT;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// One slot's front: returns 0 when no evaluation was prepared (slot idle / room finished / seed search to be continued), else
// (distinct inlier rows << 16) | distinct neighbour rows.
// ASYNC (the free-running kernel, lrg_async.inl): the slot's rows live at the fixed offset s * a.row_stride of the row arrays,
// padded to whole 32-row tiles; what other workgroups of the same launch read (rows, centre, the zeroed pooled feature) goes out
// write-through, what they wrote (the logits) is read past the L1 (lrg_fused_tile.inl, COH).
// SPEC (ASYNC only): the speculation protocol compiled in -- an instantiation of its own, so that the one-slot-per-room step keeps the register allocation it had
// without it (with both in one function the step saved and restored 32 instead of 21 registers per call).
template <bool ASYNC, bool SPEC = false>
__device__ __forceinline__ int lrg_front_greedy_slot(LrgFrontShared &SH, LrgSlot *slots, LrgRoom *rooms, int n_slots, const LrgGrowParams &prm,
                                                     const LrgFrontArgs &a, int32_t *big, const int s) {
    uint8_t *sh_flags = SH.flags;
    int *sh_tab = SH.tab, *sh_tabc = SH.tabc, *sh_tabe = SH.tabe;
    int (*sh_src)[512] = SH.src;
    float *sh_c = SH.c;
    int *red = SH.red, *sh_i = SH.i, *sh_list = SH.list, *wt_c = SH.wt_c, *wt_e = SH.wt_e, *sh_box = SH.box, *sh_off = SH.off;
    constexpr int PAD = ASYNC ? 32 : LRG_ROW_PAD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    LrgSlot *S = &slots[s];
    const int F = prm.feature_size, Ni = prm.n_inlier, Nn = prm.n_neighbor;
    // ---- round trip 1: everything addressed by the slot number alone ----
    const int room = S->room;
    int status = S->status;
    const int nc0 = S->nc, ne0 = S->ne;
    // what the stop decision and the random stream need of the slot, requested with the rest of round trip 1 (thread 0 used to walk
    // through these in dependent loads of its own while 1023 threads waited)
    const int seed0 = S->seed, restart0 = S->restart, step0 = S->step, stuck0 = S->stuck, steps_total0 = S->steps_total;
    const int sq0 = S->seq_mn[0], sq1 = S->seq_mn[1], sq2 = S->seq_mn[2], sq3 = S->seq_mx[0], sq4 = S->seq_mx[1], sq5 = S->seq_mx[2];
    int cur_seed = seed0, cur_restart = restart0, cur_step = step0, cur_target = S->target;
    bool lists_ready = false;            // a fresh seed: its index lists come from the seed search, no box query
    uint8_t *cur = S->cur;
    int32_t *cur_idx = S->cur_idx, *cand_idx = S->cand_idx;
    const int rows_off_in = a.slot_rows[4 * s + 2], rows_off_nb = a.slot_rows[4 * s + 3];
    const int tail_b_in = (ASYNC && a.tail_base) ? a.tail_base[2 * s] : -1, tail_b_nb = (ASYNC && a.tail_base) ? a.tail_base[2 * s + 1] : -1;
    // (free-running: the centre went out write-through -- read it the way other workgroups do, not through a line of this CU's L1)
    const float c0 = ASYNC ? lrg_ld_coh(a.center + s * 16 + 0) : a.center[s * 16 + 0], c1 = ASYNC ? lrg_ld_coh(a.center + s * 16 + 1) : a.center[s * 16 + 1];
    const int half = tid >> 9, j = tid & 511;                    // first half: add slot j, second half: remove slot j
    const bool mine = j < (half ? Ni : Nn);
    // which row of its set sample slot j of the LAST evaluation stands for (:240,:252): not stored by that step's sampling but worked out
    // again where it is needed -- a pure function of the slot's state, one Philox block for a padded set, and the update below waits
    // for memory, not for arithmetic
    int sj = 0;
    if (a.pooled)                    // the last evaluation's pooled feature has been consumed: zero for the next one
    {
        if constexpr (ASYNC) {      // (16 bytes per store: a dword written through is a fabric write of its own)
            if ((a.pooled_stride & 3) == 0 && (((uintptr_t)a.pooled) & 15) == 0) {
                for (int c4 = tid; c4 < (a.pooled_stride >> 2); c4 += LRG_FRONT_THREADS)
                    lrg_st_coh4(a.pooled + (long)s * a.pooled_stride, (unsigned)c4 * 16u, make_float4(0.f, 0.f, 0.f, 0.f));
            } else {
                for (int c = tid; c < a.pooled_stride; c += LRG_FRONT_THREADS) lrg_st_coh(a.pooled + (long)s * a.pooled_stride + c, 0.f);
            }
        } else {
            for (int c = tid; c < a.pooled_stride; c += LRG_FRONT_THREADS) a.pooled[(long)s * a.pooled_stride + c] = 0.f;
        }
    }
    if (room < 0) {
        if (tid == 0) { a.slot_rows[4 * s + 0] = 0; a.slot_rows[4 * s + 1] = 0; big[2 * s] = 0; }
        return 0;
    }
    // ---- round trip 2: the room ----
    LrgRoom *R = &rooms[room];
    const int n = R->n;
    const float *points = R->points;
    const int32_t *obj = R->obj_id;
    const uint32_t *pvox = R->pvox;
    uint8_t *visited = R->visited;
    const int ox = R->vox_origin[0], oy = R->vox_origin[1], oz = R->vox_origin[2];
    LrgVoxIndex VI;
    VI.grid = R->vgrid; VI.gx = R->vgrid_dim[0]; VI.gy = R->vgrid_dim[1]; VI.gz = R->vgrid_dim[2];
    VI.ox = ox; VI.oy = oy; VI.oz = oz;
    VI.hkeys = R->hash_keys; VI.hvals = R->hash_vals; VI.hmask = R->hash_mask;
    const uint32_t k0 = prm.rng_seed, k1 = (uint32_t)R->room_id;
    const int K = (ASYNC && SPEC) ? a.spec_k : 1;
    constexpr bool spec = ASYNC && SPEC;                 // several regions of this room in flight (see "speculation" above)
    const int g0 = spec ? (s / K) * K : s;
    int list_n = nc0, list_count = 0;            // the pending region's list length and member count, carried in registers through this call
    int spec_void = 0, my_pos = INT_MAX, cursor = 0;
    if (spec) {
        // the slot's and the room's words that steer this call, read once by thread 0 (see lrg_spec_is_head) and broadcast
        if (tid == 0) {
            sh_tabe[0] = lrg_ld_coh(&S->status); sh_tabe[1] = lrg_ld_coh(&S->count); sh_tabe[2] = lrg_ld_coh(&S->spec_flags);
            sh_tabe[3] = lrg_ld_coh(&S->spec_pos); sh_tabe[4] = lrg_ld_coh(&R->seed_cursor); sh_tabe[5] = lrg_ld_coh(&S->nc);
        }
        __syncthreads();
        status = sh_tabe[0]; list_count = sh_tabe[1]; spec_void = sh_tabe[2] & 1; my_pos = sh_tabe[3]; cursor = sh_tabe[4]; list_n = sh_tabe[5];
        __syncthreads();
    }
    if (spec && spec_void) {
        // the region in progress was voided by a commit of an earlier region: dropped here, at the slot's first turn since (the evaluation that was
        // in flight is ignored); its list holds every member of the mask (all of it after a box query, a superset while pending)
        if (status == LRG_ACTIVE || status == LRG_PENDING || lrg_is_stop(status)) {
            for (int i = tid; i < list_n; i += LRG_FRONT_THREADS) cur[lrg_ld_coh(&cur_idx[i])] = 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the mask is read again by other wavefronts behind the barrier below)
            if (tid == 0) {
                if (a.spec_stats) {
                    atomicAdd(&a.spec_stats[1], (unsigned long long)(steps_total0 + (status == LRG_ACTIVE ? 1 : 0)));      // (+ the evaluation that was in flight)
                    atomicAdd(&a.spec_stats[2], (unsigned long long)steps_total0);                                         // (steps counted in stats[2] that no region keeps)
                }
                LRG_SPEC_EV(a, 4, s, seed0, status, nc0);
                S->status = LRG_WAIT;                    // (spec_pos stays: the seed search below starts there)
            }
            status = LRG_WAIT;
        }
        if (tid == 0) { S->spec_flags = 0; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
    }
    const int entry_status = status;
    TRACE2(s, 0);
    long long t_phase = (ASYNC && LRG_ASYNC_DEBUG && a.phase_dbg) ? wall_clock64() : 0;
    auto phase = [&](int i) {      // ticks since the last stamp -> accumulator i (thread 0)
        if constexpr (ASYNC && LRG_ASYNC_DEBUG) if (a.phase_dbg && tid == 0) { const long long now = wall_clock64(); atomicAdd(&a.phase_dbg[i], (unsigned long long)(now - t_phase)); t_phase = now; }
    };
    const long long tick0 = a.phase_ticks ? wall_clock64() : 0;

    int q_nc = nc0, q_ne = ne0;        // sizes of the lists the sampling below draws from (set by the seed path / the box query)
    // =========================== (1) mask update of the evaluation just finished (:262-288) ===========================
    int id0[4];                      // the first 4096 entries of the old index list and their voxel words, kept for the commit
    uint32_t pv0[4];
    int al0 = 0;                     // bit k: entry k is still a member after the update
    if (status == LRG_ACTIVE) {
        // The update by the slot's own lists (round 6).  A sample slot's point is entry `pos` of the list its side was sampled from (:238-240,:250-252) -- known
        // before the evaluation, like the point's voxel word -- and the voxel the reference re-derives from the centred row, rint(((x - c) + c) / res) (:271-276),
        // is that point's own voxel unless the float32 round trip through the centre moves x across a voxel boundary.  So where no taken slot's voxel moved
        // (checked for all of them: SH.upd[0]), an add switches on the slot's own candidate -- not in the mask, by the box query (:226-228): elected among the copies
        // of its row by a flag in LDS, no atomic on the mask's word and no voxel look-up -- and a remove switches off the slot's own member and marks its list
        // position in a bitmap in LDS, from which the survivors' count and bounding box (:292-293) follow without reading the mask back.  Behind the logits' trip
        // the update is LDS work and fire-and-forget stores: four dependent memory round trips less (grid look-up, atomicOr, mask read-back, added points'
        // voxel words; 7.3 -> ~3 us of the front step).  Same mask, same lists, bit for bit; a moved voxel (rare) takes the general form below for that step.
        unsigned *bm_rm = reinterpret_cast<unsigned *>(sh_flags);        // bit i: entry i of the old index list was removed by this step
        {
            const int bm_words = (nc0 + 31) >> 5;                          // (<= 4096 words: rooms up to 131 072 points)
            for (int w = tid; w < bm_words; w += LRG_FRONT_THREADS) bm_rm[w] = 0u;
            if (tid < 512) sh_tab[tid] = 0;                                // per neighbour row: its point was switched on (copies of a row draw for themselves, :266)
        }
        if (tid == 0) { sh_i[0] = 0; sh_i[1] = 0; sh_i[5] = 0; sh_i[6] = 0; }
#pragma unroll
        for (int k = 0; k < 4; ++k) id0[k] = spec ? lrg_ld_coh(&cur_idx[min(tid + k * LRG_FRONT_THREADS, nc0 - 1)]) : cur_idx[min(tid + k * LRG_FRONT_THREADS, nc0 - 1)];
        const int nside = half ? nc0 : ne0, Nside = half ? Ni : Nn;
        if (mine && nside < Nside)
            sj = (int)lrg_sample_position((uint32_t)j, (uint32_t)nside, (uint32_t)Nside, half ? LRG_PURPOSE_INLIER : LRG_PURPOSE_NEIGHBOR,
                                          (uint32_t)seed0, (uint32_t)restart0, (uint32_t)step0, k0, k1);
        const int srow = nside < Nside ? sj : j;                                              // a padded slot reads its source row
        long row = (half ? rows_off_in : rows_off_nb) + srow;
        if constexpr (ASYNC) {      // (a row of the side's tail: in the shared rows, where the evaluation's gather put it)
            const int tb = half ? tail_b_in : tail_b_nb, first_tail = min(nside, Nside) & ~31;
            if (tb >= 0 && srow >= first_tail) row = (long)a.tail_row0 + tb + (srow - first_tail);
        }
        int idx = -1;
        int correct = 0;
        bool take = false;
        int pos = 0, own = -1;                        // the source row's position in the list it was sampled from (the gather's arithmetic), and the point there
        float px = 0.f, py = 0.f, pz = 0.f;
        if (mine) {
            // (a padded set's rows are its members in order, a full set's the prefix of the permutation)
            pos = nside < Nside ? srow : (int)lrg_sample_position((uint32_t)srow, (uint32_t)nside, (uint32_t)Nside, half ? LRG_PURPOSE_INLIER : LRG_PURPOSE_NEIGHBOR,
                                                                  (uint32_t)seed0, (uint32_t)restart0, (uint32_t)step0, k0, k1);
            const int32_t *lst = half ? cur_idx : cand_idx;
            own = spec ? lrg_ld_coh(&lst[pos]) : lst[pos];
            const float4 u = (half ? a.upd_in : a.upd_nb)[(long)s * Nside + srow];
            px = u.x; py = u.y; pz = u.z;
            const float *lgp = (half ? a.rmv_logits : a.add_logits) + 2 * row;
            float lg[2];
            if constexpr (ASYNC) { const float2 t = lrg_ld_coh2(lgp); lg[0] = t.x; lg[1] = t.y; }      // (written by a tile team of this launch)
            else { lg[0] = lgp[0]; lg[1] = lgp[1]; }
            const int gtf = u.w != 0.f;
            correct = (lg[1] > lg[0] ? 1 : 0) == gtf;                                        // add_acc / remove_acc (util:174-180)
            if (prm.policy == 2) take = gtf != 0;                                            // :268-269
            else {
                const float conf = lrg_conf(lg);                                             // :262-263
                if (prm.policy == 1) take = conf > 0.5f;                                     // :264-265
                else take = lrg_uniform01(lrg_rng_word((uint32_t)j, half ? LRG_PURPOSE_RMV : LRG_PURPOSE_ADD, (uint32_t)seed0,
                                                       (uint32_t)restart0, (uint32_t)step0, k0, k1)) < conf;   // :266-267
            }
#ifdef LRG_SPEC_TRACE
            if (spec && j == 0) LRG_SPEC_EV(a, 5 + half, s, seed0, ((long long)__float_as_uint(lg[0]) << 32) | __float_as_uint(lg[1]), ((long long)take << 40) | ((long long)(row & 0xFFFFF) << 20) | (srow & 0xFFFFF));
#endif
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) pv0[k] = pvox[id0[k]];
        const uint32_t pv_own = mine ? pvox[max(own, 0)] : 0u;                                 // (the slot's own point's voxel word: the test below, an added point's share of the new bounding box)
        // the voxel the reference looks up: the rows are stored uncentred, so (x - c) + c in float32, as it centres (:243,:246) and un-centres (:271,:275) the row
        int vx = 0, vy = 0, vz = 0;
        if (mine && take) {
            vx = lrg_voxel_of(__fadd_rn(__fsub_rn(px, c0), c0), prm.resolution);              // :271-272 / :275-276
            vy = lrg_voxel_of(__fadd_rn(__fsub_rn(py, c1), c1), prm.resolution);
            vz = lrg_voxel_of(pz, prm.resolution);
#ifndef LRG_UPDATE_GENERAL      // (build switch: every update in the general form, as up to round 5)
            if (vx - ox != LRG_PVX(pv_own) || vy - oy != LRG_PVY(pv_own) || vz - oz != LRG_PVZ(pv_own)) SH.upd[0] = 1;      // (whoever stores, stores 1; the room's grid holds `own` in that voxel)
#else
            SH.upd[0] = 1;
#endif
        }
        {
            const int wsum = lrg_wave_sum_i32(correct);          // a wavefront lies in one half (512 = 8 wavefronts)
            __syncthreads();
            if (lane == 0 && wsum) atomicAdd(&sh_i[5 + half], wsum);
        }
        const bool by_lists = SH.upd[0] == 0;                     // (workgroup-uniform: written before the barrier, cleared behind the next ones)
        int cnt = 0;
        int mn0 = INT_MAX, mn1 = INT_MAX, mn2 = INT_MAX, mx0 = INT_MIN, mx1 = INT_MIN, mx2 = INT_MIN;
        if (by_lists) {
            if (mine && take) {
                if (!half) {                                                                  // :283-285
                    if (atomicOr(&sh_tab[srow], 1) == 0) {
                        cur[own] = 1;
                        sh_i[0] = 1;
                        const int k = atomicAdd(&sh_i[1], 1);
                        sh_src[0][k] = own; sh_src[1][k] = (int)pv_own;
                    }
                } else {                                                                      // :286-287
                    atomicOr(&bm_rm[pos >> 5], 1u << (pos & 31));
                    cur[own] = 0;
                }
            }
            __syncthreads();
            // members and bounding box of the new mask (:292-293): the old members whose list position was not removed + the points switched on
            // (a candidate is no member: nothing is added and removed in the same step here)
            const int nadd = sh_i[1];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = tid + k * LRG_FRONT_THREADS;
                if (i < nc0 && !((bm_rm[i >> 5] >> (i & 31)) & 1u)) {
                    al0 |= 1 << k;
                    ++cnt;
                    const int x = LRG_PVX(pv0[k]), y = LRG_PVY(pv0[k]), z = LRG_PVZ(pv0[k]);
                    mn0 = min(mn0, x); mn1 = min(mn1, y); mn2 = min(mn2, z);
                    mx0 = max(mx0, x); mx1 = max(mx1, y); mx2 = max(mx2, z);
                }
            }
            for (int i0 = 4 * LRG_FRONT_THREADS; i0 < nc0; i0 += 4 * LRG_FRONT_THREADS) {       // regions above 4096 points
                int id[4];
                uint32_t pv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) id[k] = cur_idx[min(i0 + k * LRG_FRONT_THREADS + tid, nc0 - 1)];
#pragma unroll
                for (int k = 0; k < 4; ++k) pv[k] = pvox[id[k]];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * LRG_FRONT_THREADS + tid;
                    if (i < nc0 && !((bm_rm[i >> 5] >> (i & 31)) & 1u)) {
                        ++cnt;
                        const int x = LRG_PVX(pv[k]), y = LRG_PVY(pv[k]), z = LRG_PVZ(pv[k]);
                        mn0 = min(mn0, x); mn1 = min(mn1, y); mn2 = min(mn2, z);
                        mx0 = max(mx0, x); mx1 = max(mx1, y); mx2 = max(mx2, z);
                    }
                }
            }
            for (int k = tid; k < nadd; k += LRG_FRONT_THREADS) {
                const uint32_t pv = (uint32_t)sh_src[1][k];
                ++cnt;
                const int x = LRG_PVX(pv), y = LRG_PVY(pv), z = LRG_PVZ(pv);
                mn0 = min(mn0, x); mn1 = min(mn1, y); mn2 = min(mn2, z);
                mx0 = max(mx0, x); mx1 = max(mx1, y); mx2 = max(mx2, z);
            }
        } else {
        // ---- the general form: the voxel's point from the room's grid / hash, a word-wide atomicOr electing the slot that switches a point on, the mask read back ----
        if (mine && take) idx = lrg_voxel_index(VI, vx, vy, vz);
        if (!half && idx >= 0) {                                                             // :283-285
            unsigned *w = reinterpret_cast<unsigned *>(cur + (idx & ~3));
            const unsigned bit = 1u << (8 * (idx & 3));
            if (!(atomicOr(w, bit) & bit)) { sh_i[0] = 1; sh_src[0][atomicAdd(&sh_i[1], 1)] = idx; }
        }
        __syncthreads();
        if (half && idx >= 0) cur[idx] = 0;                                                  // :286-287
        __syncthreads();
        // members and bounding box of the new mask (:292-293): surviving old members + surviving new ones
        const int nadd = sh_i[1];
        {
            int al[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) al[k] = cur[id0[k]];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tid + k * LRG_FRONT_THREADS < nc0 && al[k]) {
                    al0 |= 1 << k;
                    ++cnt;
                    const int x = LRG_PVX(pv0[k]), y = LRG_PVY(pv0[k]), z = LRG_PVZ(pv0[k]);
                    mn0 = min(mn0, x); mn1 = min(mn1, y); mn2 = min(mn2, z);
                    mx0 = max(mx0, x); mx1 = max(mx1, y); mx2 = max(mx2, z);
                }
        }
        for (int i0 = 4 * LRG_FRONT_THREADS; i0 < nc0; i0 += 4 * LRG_FRONT_THREADS) {       // regions above 4096 points
            int id[4], al[4];
            uint32_t pv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) id[k] = cur_idx[min(i0 + k * LRG_FRONT_THREADS + tid, nc0 - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k) { al[k] = cur[id[k]]; pv[k] = pvox[id[k]]; }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k * LRG_FRONT_THREADS + tid < nc0 && al[k]) {
                    ++cnt;
                    const int x = LRG_PVX(pv[k]), y = LRG_PVY(pv[k]), z = LRG_PVZ(pv[k]);
                    mn0 = min(mn0, x); mn1 = min(mn1, y); mn2 = min(mn2, z);
                    mx0 = max(mx0, x); mx1 = max(mx1, y); mx2 = max(mx2, z);
                }
        }
        for (int k = tid; k < nadd; k += LRG_FRONT_THREADS) {
            const int idx2 = sh_src[0][k];
            const uint32_t pv = pvox[idx2];
            if (cur[idx2]) {                                          // (a point added and removed in the same step stays out)
                ++cnt;
                const int x = LRG_PVX(pv), y = LRG_PVY(pv), z = LRG_PVZ(pv);
                mn0 = min(mn0, x); mn1 = min(mn1, y); mn2 = min(mn2, z);
                mx0 = max(mx0, x); mx1 = max(mx1, y); mx2 = max(mx2, z);
            }
        }
        }      // (general form)
        lrg_block_bbox(cnt, mn0, mn1, mn2, mx0, mx1, mx2, red);
        if (tid == 0) {
            const int upd = sh_i[0];
            SH.upd[0] = 0;                       // (read by everybody before the barriers of the reduction above)
            S->pad = 0;
            S->step = step0 + 1;
            S->steps_total = steps_total0 + 1;                                               // :288
            S->acc_add = sh_i[5]; S->acc_rmv = sh_i[6];
            if (a.stats) atomicAdd(reinterpret_cast<unsigned long long *>(&a.stats[2]), 1ULL);
            // lrg_stop_logic (:291-306) on the values at hand: the same decisions and the same stores, without its loads
            S->scan_cnt = 0;
            S->scan_mn[0] = S->scan_mn[1] = S->scan_mn[2] = INT_MAX;
            S->scan_mx[0] = S->scan_mx[1] = S->scan_mx[2] = INT_MIN;
            S->updated = -1;
            S->count = cnt;
            int st = LRG_ACTIVE;
            if (!upd) st = LRG_STOP_NOEXPAND;                                                // :304-306
            else if (cnt == 0) st = LRG_STOP_EMPTY;                                          // :292 would raise
            else {
                const int a0 = mn0 + ox, a1 = mn1 + oy, a2 = mn2 + oz, b0 = mx0 + ox, b1 = mx1 + oy, b2 = mx2 + oz;
                S->mn[0] = a0; S->mn[1] = a1; S->mn[2] = a2; S->mx[0] = b0; S->mx[1] = b1; S->mx[2] = b2;     // :292-293
                sh_box[0] = a0; sh_box[1] = a1; sh_box[2] = a2; sh_box[3] = b0; sh_box[4] = b1; sh_box[5] = b2;
                const bool grew = a0 < sq0 || a1 < sq1 || a2 < sq2 || b0 > sq3 || b1 > sq4 || b2 > sq5;        // :294
                if (!grew && stuck0 >= 1) st = LRG_STOP_STUCK;                               // :295-297
                else {
                    S->stuck = grew ? 0 : stuck0 + 1;                                        // :299,:301
                    S->seq_mn[0] = min(sq0, a0); S->seq_mn[1] = min(sq1, a1); S->seq_mn[2] = min(sq2, a2);
                    S->seq_mx[0] = max(sq3, b0); S->seq_mx[1] = max(sq4, b1); S->seq_mx[2] = max(sq5, b2);
                }
            }
            if (st != LRG_ACTIVE) { S->status = st; S->last_reason = st; }
            if (spec) LRG_SPEC_EV(a, 2, s, seed0, (cnt << 16) | (sh_i[1] << 4) | st, (nc0 << 16) | ne0);
            sh_i[2] = st;
            sh_box[6] = cnt;
        }
        cur_step = step0 + 1;
        LRG_LDS_BARRIER();           // (status, box and count go through LDS; nobody reads thread 0's stores to the slot before the next full barrier)
        status = sh_i[2];
    }
    TRACE2(s, 1); phase(1);

    // =========================== (2) commit (:210-217), next seed (:186-188), reset (:197-204) ===========================
    if (spec) {
        if (lrg_is_stop(status)) {
            // the region has stopped; it is committed in its turn (lrg_spec_commit) from its index list: after an update the old entries (those
            // the update removed are skipped there) + the points this step switched on; after a stop taken by the box query the list as it is
            if (entry_status == LRG_ACTIVE) {
                const int nadd = sh_i[1];
                for (int k = tid; k < nadd; k += LRG_FRONT_THREADS) cur_idx[nc0 + k] = sh_src[0][k];
                if (tid == 0) S->nc = nc0 + nadd;
                list_n = nc0 + nadd;
                list_count = sh_box[6];
            }
            if (tid == 0) S->status = LRG_PENDING;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (list and length are read by the commit, by every wavefront)
            __syncthreads();
            status = LRG_PENDING;
        }
        const int32_t *order = R->order;
        // (what this call itself changes is carried in registers, not read back: a barrier does not wait for thread 0's global stores)
        for (int round = 0; status == LRG_PENDING || status == LRG_WAIT; ++round) {
            if (status == LRG_PENDING) {
                if (!lrg_spec_is_head(SH, slots, g0, K, s, my_pos)) {      // an earlier seed's region is still in flight: wait
                    if (tid == 0) { a.slot_rows[4 * s + 0] = 0; a.slot_rows[4 * s + 1] = 0; big[2 * s] = 0; }
                    return 0;
                }
                lrg_spec_commit(SH, slots, R, g0, K, s, prm, a, list_n, list_count);
                status = LRG_WAIT;
                my_pos = INT_MAX;
            }
            if (round >= LRG_SEED_TRIES) {          // (a run of one-point regions: the next turn goes on)
                if (tid == 0) { a.slot_rows[4 * s + 0] = 0; a.slot_rows[4 * s + 1] = 0; big[2 * s] = 0; }
                return 0;
            }
            // the next seed: a voided region's own, while unvisited; else the next unvisited position of the order that no slot holds (:186-188)
            const int own = my_pos;
            int found = -1;
            if (own != INT_MAX) {                // (the voided region's own seed: still unvisited? -- thread 0's look, broadcast)
                __syncthreads();
                if (tid == 0) sh_i[3] = __hip_atomic_load(&visited[order[own]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0 : 1;
                __syncthreads();
                if (sh_i[3]) found = own;
                __syncthreads();
            }
            if (found < 0) {
                while (cursor < n) {
                    const int pos = cursor + tid;
                    int cand = INT_MAX;
                    if (pos < n && !visited[order[pos]]) cand = pos;
                    if (tid == 0) sh_i[3] = INT_MAX;
                    __syncthreads();
                    if (cand != INT_MAX) atomicMin(&sh_i[3], cand);
                    __syncthreads();
                    const int best = sh_i[3];
                    __syncthreads();
                    if (best != INT_MAX) { found = best; break; }
                    cursor += LRG_FRONT_THREADS;
                }
                if (found >= 0) cursor = found + 1;
                if (tid == 0) R->seed_cursor = found >= 0 ? cursor : n;
            }
            if (found < 0) {
                // no seed left for this slot.  The room is finished when that holds for every slot of the group: the last one reports it.
                if (tid == 0) {
                    bool all = true;
                    for (int t = g0; t < g0 + K; ++t)
                        if (t != s && slots[t].status != LRG_IDLE) all = false;
                    S->spec_pos = INT_MAX; S->seed = -1; S->count = -1;
                    if (all) {
                        R->done = 1;
                        S->status = LRG_DONE;
                        if (a.stats) {
                            unsigned long long k = atomicAdd(reinterpret_cast<unsigned long long *>(&a.stats[1]), 1ULL);
                            a.stats[4 + (k % LRG_DONE_RING)] = (int64_t)s | (a.fill_in_launch ? (int64_t)1 << 31 : 0) | ((int64_t)room << 32);
                        }
                    } else {
                        S->status = LRG_IDLE;                // (room still bound: the group is rebound as a whole)
                    }
                    a.slot_rows[4 * s + 0] = 0; a.slot_rows[4 * s + 1] = 0; big[2 * s] = 0;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the serving loop reads the status right behind this call)
                }
                __syncthreads();
                return 0;
            }
            const int sd = order[found];
            if (tid < 64) {
                int idx = -1;
                if (tid < 27 && tid != 13) {
                    const int dx = tid / 9 - 1, dy = (tid / 3) % 3 - 1, dz = tid % 3 - 1;
                    const int32_t *v = R->voxels + 3 * (long)sd;
                    idx = lrg_voxel_index(VI, v[0] + dx, v[1] + dy, v[2] + dz);
                    if (idx >= 0 && (visited[idx] || idx == sd)) idx = -1;
                }
                int rank = 0, total = 0;
                for (int l = 0; l < 27; ++l) {
                    const int o = __shfl(idx, l);
                    if (o >= 0) { ++total; if (idx >= 0 && o < idx) ++rank; }
                }
                if (idx >= 0) sh_list[rank] = idx;
                if (tid == 0) sh_i[4] = total;
            }
            __syncthreads();
            const int total = sh_i[4];
            __syncthreads();
            if ((int)tid < total) cand_idx[tid] = sh_list[tid];
            if (tid == 0) {
                // (the slot's mask is all zero here: a commit and a drop both clear their members)
                cur[sd] = 1;                                                                  // :197-198
                cur_idx[0] = sd;
                S->seed = sd; S->restart = 0; S->step = 0; S->stuck = 0; S->updated = -1;
                S->nc = 1; S->ne = total; S->count = 1;
                for (int d = 0; d < 3; ++d) {
                    const int v = R->voxels[3 * sd + d];
                    S->mn[d] = v; S->mx[d] = v; S->seq_mn[d] = v; S->seq_mx[d] = v;          // :199-202
                }
                const int tg = obj ? obj[sd] : 0;
                S->target = tg;
                sh_box[7] = tg;
                S->pad = 1;
                S->scan_cnt = 0;
                S->scan_mn[0] = S->scan_mn[1] = S->scan_mn[2] = INT_MAX;
                S->scan_mx[0] = S->scan_mx[1] = S->scan_mx[2] = INT_MIN;
                S->steps_total = 0; S->best_count = -1; S->best_restart = INT_MAX;
                S->acc_add = -1; S->acc_rmv = -1;
                S->spec_pos = found;
                LRG_SPEC_EV(a, 1, s, sd, found, (total << 16) | (total > 0 ? (sh_list[0] & 0xFFFF) : 0xFFFF));
                // a seed without an unvisited neighbour is a region of its own, 'noneighbor' before any step (:233-235): committed in its turn like every other
                S->last_reason = total > 0 ? 0 : LRG_STOP_NONEIGHBOR;
                S->status = total > 0 ? LRG_ACTIVE : LRG_PENDING;
            }
            my_pos = found;
            list_n = 1; list_count = 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (mask, lists and slot fields are read by other wavefronts behind this barrier: the sampling, the medians, a commit)
            __syncthreads();
            if (total > 0) {
                q_nc = 1; q_ne = total;
                cur_seed = sd; cur_restart = 0; cur_step = 0;
                lists_ready = true;
                status = LRG_ACTIVE;
                cur_target = sh_box[7];
            } else {
                status = LRG_PENDING;
            }
        }
    } else {
    if (lrg_is_stop(status)) {
        // members of the finished region: after an update, the survivors found above; after a stop taken by the box query
        // ('noneighbor', step cap), the list that query compacted (every entry a member)
        const bool from_update = entry_status == LRG_ACTIVE;
        const int count = from_update ? sh_box[6] : S->count;
        const int labeled = count > prm.cluster_threshold;                                   // :213
        const int cid = R->next_cluster_id;
        int32_t *label = R->label;
        if (from_update) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (al0 >> k & 1) { visited[id0[k]] = 1; if (labeled) label[id0[k]] = cid; cur[id0[k]] = 0; }   // :212,:214 + reset
            for (int i = 4 * LRG_FRONT_THREADS + tid; i < nc0; i += LRG_FRONT_THREADS) {
                const int id = cur_idx[i];
                if (cur[id]) { visited[id] = 1; if (labeled) label[id] = cid; cur[id] = 0; }
            }
            const int nadd = sh_i[1];
            for (int k = tid; k < nadd; k += LRG_FRONT_THREADS) {
                const int id = sh_src[0][k];
                if (cur[id]) { visited[id] = 1; if (labeled) label[id] = cid; cur[id] = 0; }
            }
        } else {
            for (int i = tid; i < nc0; i += LRG_FRONT_THREADS) {
                const int id = cur_idx[i];
                visited[id] = 1; if (labeled) label[id] = cid; cur[id] = 0;
            }
        }
        if (tid == 0) {
            int32_t *log = R->region_log + LRG_LOG_WORDS * (long)R->n_regions;
            log[0] = S->seed; log[1] = S->steps_total; log[2] = count; log[3] = S->last_reason; log[4] = labeled; log[5] = 0;
            log[6] = S->acc_add; log[7] = S->acc_rmv;
            R->n_regions += 1;
            if (labeled) R->next_cluster_id = cid + 1;                                       // :215
            if (a.stats) atomicAdd(reinterpret_cast<unsigned long long *>(&a.stats[0]), 1ULL);
        }
        __syncthreads();
        status = LRG_WAIT;
    }
    if (status == LRG_WAIT) {
        // next unvisited seed in curvature order; its first box query (a one-voxel box) is answered from the voxel hash, and a
        // seed without neighbours is committed on the spot (see lrg_advance_group)
        int cursor = R->seed_cursor;
        int seed = -1, n_cand = 0;
        const int32_t *order = R->order;
        for (int tries = 0; tries < LRG_SEED_TRIES; ++tries) {
            int found = -1;
            while (cursor < n) {
                const int pos = cursor + tid;
                int cand = INT_MAX;
                if (pos < n && !visited[order[pos]]) cand = pos;
                if (tid == 0) sh_i[3] = INT_MAX;
                __syncthreads();
                if (cand != INT_MAX) atomicMin(&sh_i[3], cand);
                __syncthreads();
                const int best = sh_i[3];
                __syncthreads();
                if (best != INT_MAX) { found = best; break; }
                cursor += LRG_FRONT_THREADS;
            }
            if (found < 0) {
                if (tid == 0) {
                    R->seed_cursor = n;
                    R->done = 1;
                    S->status = LRG_DONE;
                    if (a.stats) {
                        unsigned long long k = atomicAdd(reinterpret_cast<unsigned long long *>(&a.stats[1]), 1ULL);
                        a.stats[4 + (k % LRG_DONE_RING)] = (int64_t)s | (a.fill_in_launch ? (int64_t)1 << 31 : 0) | ((int64_t)room << 32);      // (slot, room: a free-running launch may have rebound the slot by the time the host looks)
                    }
                    a.slot_rows[4 * s + 0] = 0; a.slot_rows[4 * s + 1] = 0; big[2 * s] = 0;
                }
                        return 0;
            }
            const int sd = order[found];
            cursor = found + 1;
            if (tid < 64) {
                int idx = -1;
                if (tid < 27 && tid != 13) {
                    const int dx = tid / 9 - 1, dy = (tid / 3) % 3 - 1, dz = tid % 3 - 1;
                    const int32_t *v = R->voxels + 3 * (long)sd;
                    idx = lrg_voxel_index(VI, v[0] + dx, v[1] + dy, v[2] + dz);
                    if (idx >= 0 && (visited[idx] || idx == sd)) idx = -1;
                }
                int rank = 0, total = 0;
                for (int l = 0; l < 27; ++l) {
                    const int o = __shfl(idx, l);
                    if (o >= 0) { ++total; if (idx >= 0 && o < idx) ++rank; }
                }
                if (idx >= 0) sh_list[rank] = idx;
                if (tid == 0) sh_i[4] = total;
            }
            __syncthreads();
            const int total = sh_i[4];
            __syncthreads();
            if (total > 0) { seed = sd; n_cand = total; break; }
            if (tid == 0) {                                          // no neighbour: the region is the seed alone (:233-235)
                const int labeled = 1 > prm.cluster_threshold;
                visited[sd] = 1;
                if (labeled) { R->label[sd] = R->next_cluster_id; R->next_cluster_id += 1; }
                int32_t *log = R->region_log + LRG_LOG_WORDS * (long)R->n_regions;
                log[0] = sd; log[1] = 0; log[2] = 1; log[3] = LRG_STOP_NONEIGHBOR; log[4] = labeled; log[5] = 0; log[6] = -1; log[7] = -1;
                R->n_regions += 1;
                if (a.stats) atomicAdd(reinterpret_cast<unsigned long long *>(&a.stats[0]), 1ULL);
            }
            __syncthreads();
        }
        if (tid == 0) R->seed_cursor = cursor;
        if (seed < 0) {              // try budget spent on isolated points: the next call continues the search
            if (tid == 0) {
                S->seed = -1; S->status = LRG_WAIT; S->count = -1;
                a.slot_rows[4 * s + 0] = 0; a.slot_rows[4 * s + 1] = 0; big[2 * s] = 0;
            }
                return 0;
        }
        // the slot's mask is all zero here (every region clears its members at commit; the host zeroes it when binding)
        if ((int)tid < n_cand) cand_idx[tid] = sh_list[tid];
        q_nc = 1; q_ne = n_cand;
        cur_seed = seed; cur_restart = 0; cur_step = 0;
        lists_ready = true;
        if (tid == 0) {
            cur[seed] = 1;                                                                   // :197-198
            cur_idx[0] = seed;
            S->seed = seed; S->restart = 0; S->step = 0; S->stuck = 0; S->updated = -1;
            S->nc = 1; S->ne = n_cand; S->count = 1;
            for (int d = 0; d < 3; ++d) {
                const int v = R->voxels[3 * seed + d];
                S->mn[d] = v; S->mx[d] = v; S->seq_mn[d] = v; S->seq_mx[d] = v;             // :199-202
            }
            const int tg = obj ? obj[seed] : 0;
            S->target = tg;
            sh_box[7] = tg;
            S->pad = 1;                                                                      // lists ready
            S->scan_cnt = 0;
            S->scan_mn[0] = S->scan_mn[1] = S->scan_mn[2] = INT_MAX;
            S->scan_mx[0] = S->scan_mx[1] = S->scan_mx[2] = INT_MIN;
            S->steps_total = 0; S->best_count = -1; S->best_restart = INT_MAX; S->last_reason = 0;
            S->acc_add = -1; S->acc_rmv = -1;
            S->status = LRG_ACTIVE;
        }
        __syncthreads();
        status = LRG_ACTIVE;
        cur_target = sh_box[7];
    }
    }      // (not speculating)
    TRACE2(s, 2); phase(2);
    const long long tick1 = a.phase_ticks ? wall_clock64() : 0;
    if (a.phase_ticks && tid == 0) a.phase_ticks[2 * s + 0] += tick1 - tick0;
    if (status != LRG_ACTIVE) {      // DONE / IDLE
        if (tid == 0) { a.slot_rows[4 * s + 0] = 0; a.slot_rows[4 * s + 1] = 0; big[2 * s] = 0; }
        return 0;
    }

    // =========================== (3) dilated voxel-box query with ordered compaction (:221-235) ===========================
    // (round 6: where the grid query's two index lists are short -- <= 2048 entries each, the bitmaps in the first half of the flag bytes -- they are ALSO kept in LDS
    //  behind the bitmaps, for the sampling and the wave medians of this same step: a trip to L2 less in each)
    bool lds_lists = false;
    int *lds_cur = reinterpret_cast<int *>(sh_flags + 16384), *lds_cand = lds_cur + 2048;
    if (!lists_ready) {
        const int nchunk = (n + LRG_SCAN_CHUNK - 1) / LRG_SCAN_CHUNK;
        const int lo0 = max(sh_box[0] - 1 - ox, 0), lo1 = max(sh_box[1] - 1 - oy, 0), lo2 = max(sh_box[2] - 1 - oz, 0);     // :222-225
        const int hi0 = sh_box[3] + 1 - ox, hi1 = sh_box[4] + 1 - oy, hi2 = sh_box[5] + 1 - oz;
        const int ilast = (n - 1) & ~3;
        int totc = 0, tote = 0;
        // A box of up to LRG_GRID_QUERY_CELLS voxels is answered from the room's dense voxel grid: the cells of the box (rows of
        // consecutive ints along x: one round trip), the mask and visited bytes of the points found (a second one), two bitmaps
        // over the room's indices in LDS, and the ordered lists from their popcounts -- O(box) instead of O(room), and the same
        // lists: every point of the mask lies inside its bounding box, every candidate inside the dilated one (:221-229).
        const int g0 = min(hi0, VI.gx - 1), g1 = min(hi1, VI.gy - 1), g2 = min(hi2, VI.gz - 1);
        const int bx = g0 - lo0 + 1, by = g1 - lo1 + 1, bz = g2 - lo2 + 1;
        const bool by_grid = VI.grid && bx > 0 && by > 0 && bz > 0 && (long)bx * by * bz <= LRG_GRID_QUERY_CELLS;   // (workgroup-uniform)
        if (by_grid) {
            const int V = bx * by * bz;
            const int nwords = (n + 31) >> 5;                                   // <= 4096: two bitmaps = sh_flags
            unsigned *bm_c = reinterpret_cast<unsigned *>(sh_flags), *bm_e = bm_c + nwords;
            for (int w = tid; w < 2 * nwords; w += LRG_FRONT_THREADS) bm_c[w] = 0u;
            __syncthreads();
            const float rbx = 1.0f / (float)bx, rby = 1.0f / (float)by;
            constexpr int GU = 8;                                               // cells per thread and trip
            for (int v0 = 0; v0 < V; v0 += GU * LRG_FRONT_THREADS) {
                int id[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    if (u >= 4 && v0 + u * LRG_FRONT_THREADS >= V) { id[u] = -1; continue; }      // (uniform: small boxes issue 4 loads, not 16)
                    const int v = v0 + u * LRG_FRONT_THREADS + tid;
                    id[u] = -1;
                    if (v < V) {
                        const int t = lrg_div_small(v, bx, rbx), x = v - t * bx;
                        const int z = lrg_div_small(t, by, rby), y = t - z * by;
                        id[u] = VI.grid[((long)(lo2 + z) * VI.gy + (lo1 + y)) * VI.gx + lo0 + x];
                    }
                }
                int cf[GU], vf[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    if (u >= 4 && v0 + u * LRG_FRONT_THREADS >= V) { cf[u] = 0; vf[u] = 0; continue; }
                    const int i = max(id[u], 0); cf[u] = cur[i]; vf[u] = visited[i];
                }
#pragma unroll
                for (int u = 0; u < GU; ++u)
                    if (id[u] >= 0) {
                        if (cf[u]) atomicOr(&bm_c[id[u] >> 5], 1u << (id[u] & 31));
                        else if (!vf[u]) atomicOr(&bm_e[id[u] >> 5], 1u << (id[u] & 31));                                 // :226-228
                    }
            }
            __syncthreads();
            // thread t owns the words [t * wpt, (t + 1) * wpt): counts, a wavefront scan, the wavefronts' offsets, the indices
            const int wpt = (nwords + LRG_FRONT_THREADS - 1) / LRG_FRONT_THREADS;     // 1 .. 4
            unsigned mc[4], me[4];
            int pcn = 0, pen = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int w = tid * wpt + q;
                const bool in = q < wpt && w < nwords;
                mc[q] = in ? bm_c[w] : 0u; me[q] = in ? bm_e[w] : 0u;
                pcn += __popc(mc[q]); pen += __popc(me[q]);
            }
            const int packed = pcn | (pen << 16);                                     // a wavefront covers <= 8192 points
            const int incl = lrg_wave_incl_scan_i32(packed);
            if (lane == 63) { wt_c[wave] = incl & 0xFFFF; wt_e[wave] = (int)((unsigned)incl >> 16); }
            __syncthreads();
            int oc = 0, oe = 0;
#pragma unroll
            for (int w = 0; w < LRG_FRONT_THREADS / 64; ++w) {
                if (w < wave) { oc += wt_c[w]; oe += wt_e[w]; }
                totc += wt_c[w]; tote += wt_e[w];
            }
            int pc = oc + (incl & 0xFFFF) - pcn, pe = oe + (int)((unsigned)incl >> 16) - pen;
            lds_lists = nwords <= 2048 && totc <= 2048 && tote <= 2048;       // (workgroup-uniform)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int base = (tid * wpt + q) << 5;
                for (unsigned m = mc[q]; m; m &= m - 1) { const int v = base + __ffs((int)m) - 1; if (lds_lists) lds_cur[pc] = v; cur_idx[pc++] = v; }
                for (unsigned m = me[q]; m; m &= m - 1) { const int v = base + __ffs((int)m) - 1; if (lds_lists) lds_cand[pe] = v; cand_idx[pe++] = v; }
            }
        } else {
        // LRG_QUERY_TRIP chunks per trip (4 points per thread and chunk: 3 loads per 4 points), all loads of a trip in flight
        // together.  Measured with 8 (one trip up to 32 k points): no change -- the phase grows with the room through the
        // per-chunk flag / scan / compaction work (11 k cycles up to 10 k points, 24 k above 20 k), not through its round trips.
        for (int cb = 0; cb < nchunk; cb += LRG_QUERY_TRIP) {
            unsigned cw[LRG_QUERY_TRIP], vw[LRG_QUERY_TRIP];
            uint4 pw[LRG_QUERY_TRIP];
#pragma unroll
            for (int q = 0; q < LRG_QUERY_TRIP; ++q) {
                if (q >= 4 && cb + q >= nchunk) continue;        // small rooms: no loads for chunks they do not have
                const int i = min((cb + q) * LRG_SCAN_CHUNK + 4 * tid, ilast);
                cw[q] = *reinterpret_cast<const unsigned *>(cur + i);
                vw[q] = *reinterpret_cast<const unsigned *>(visited + i);
                pw[q] = *reinterpret_cast<const uint4 *>(pvox + i);
            }
#pragma unroll
            for (int q = 0; q < LRG_QUERY_TRIP; ++q) {
                const int c = cb + q;
                if (c < nchunk) {                               // workgroup-uniform
                    const int ib = c * LRG_SCAN_CHUNK + 4 * tid;
                    const unsigned pq[4] = {pw[q].x, pw[q].y, pw[q].z, pw[q].w};
                    int fc = 0, fe = 0;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        if (ib + kk < n) {
                            const int x = LRG_PVX(pq[kk]), y = LRG_PVY(pq[kk]), z = LRG_PVZ(pq[kk]);
                            if (cw[q] >> (8 * kk) & 0xFF) fc |= 1 << kk;
                            else if (!(vw[q] >> (8 * kk) & 0xFF) && x >= lo0 && x <= hi0 && y >= lo1 && y <= hi1 && z >= lo2 && z <= hi2)
                                fe |= 1 << kk;                                               // :226-228
                        }
                    }
                    // a room's points come object after object, so a region and its neighbours sit in few (chunk, wavefront)
                    // pairs: the empty ones skip the flag store and the reduction here, the scan and the stores below
                    int packed = 0;
                    if (__ballot((fc | fe) != 0)) {                                            // (wavefront-uniform)
                        sh_flags[c * LRG_FRONT_THREADS + tid] = (uint8_t)(fc | (fe << 4));
                        packed = lrg_wave_sum_i32(__popc(fc) | (__popc(fe) << 16));
                    }
                    if (lane == 0) sh_tab[c * 16 + wave] = packed;
                }
            }
        }
        __syncthreads();
        const int nent = nchunk * 16;
        int vcn = 0, ven = 0;
        if (tid < 512 && tid < nent) { const int p = sh_tab[tid]; vcn = p & 0xFFFF; ven = (int)((unsigned)p >> 16); }
        const int ic = lrg_wave_incl_scan_i32(vcn), ie = lrg_wave_incl_scan_i32(ven);
        if (tid < 512 && lane == 63) { wt_c[wave] = ic; wt_e[wave] = ie; }
        __syncthreads();
        {
            int oc = 0, oe = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                if (w < wave) { oc += wt_c[w]; oe += wt_e[w]; }
                totc += wt_c[w]; tote += wt_e[w];
            }
            if (tid < 512) { sh_tabc[tid] = oc + ic - vcn; sh_tabe[tid] = oe + ie - ven; }
        }
        __syncthreads();
        for (int c = 0; c < nchunk; ++c) {
            if (sh_tab[c * 16 + wave] == 0) continue;
            const int f = sh_flags[c * LRG_FRONT_THREADS + tid];
            const int fc = f & 15, fe = f >> 4;
            const int mine2 = __popc(fc) | (__popc(fe) << 16);
            const int excl = lrg_wave_incl_scan_i32(mine2) - mine2;
            int pc = sh_tabc[c * 16 + wave] + (excl & 0xFFFF), pe = sh_tabe[c * 16 + wave] + (int)((unsigned)excl >> 16);
            const int ib = c * LRG_SCAN_CHUNK + 4 * tid;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (fc >> k & 1) cur_idx[pc++] = ib + k;
                if (fe >> k & 1) cand_idx[pe++] = ib + k;
            }
        }
        }      // (room-wide pass)
        q_nc = totc; q_ne = tote;
        if (tid == 0) {
            S->pad = 1;
            S->nc = totc;
            S->ne = tote;
            int st = LRG_ACTIVE;
            if (tote == 0) st = LRG_STOP_NONEIGHBOR;                                         // :233-235
            else if (prm.max_region_steps > 0 && cur_step >= prm.max_region_steps) st = LRG_STOP_MAXSTEPS;
            if (st != LRG_ACTIVE) { S->status = st; S->last_reason = st; S->count = totc; }
            sh_i[2] = st;
        }
        __syncthreads();
        if (sh_i[2] != LRG_ACTIVE) {
            if (tid == 0) { a.slot_rows[4 * s + 0] = 0; a.slot_rows[4 * s + 1] = 0; big[2 * s] = 0; }
                return 0;
        }
    }
    TRACE2(s, 3); phase(3);
    // (free-running: up to 1024 points one wavefront per channel with the keys in registers, beside the gather; above, the radix select
    //  of all nine channels by the whole workgroup after it)
    const int small_max = (ASYNC && a.own_medians) ? 1024 : LRG_FRONT_SMALL;

    // =========================== (4) sampling (:237-252), centre (:241), gather (:242-254) ===========================
    const int nc = q_nc, ne = q_ne;                      // (known to every thread: no trip through S->nc / S->ne)
    const int rin = min(nc, Ni), rnb = min(ne, Nn);
    const bool is_big = nc > small_max;
    int oi = 0, on = 0;
    if constexpr (ASYNC) {
        if (tid < 2) {                                   // (lane 0: the inlier side, lane 1: the neighbour side -- one round trip for both)
            // the rows beyond the side's last full tile: reserved in the shared rows (the next slots' tails follow: one tile for several of them).  Out of
            // shared rows (a launch longer than they were sized for): the side pads a tile of its own, as without them.
            int tb = -1, dead = 0;
            const int tl = (tid ? rnb : rin) & 31;
            if (a.tail_cur && a.rows16 && tl) {
                const int b = __hip_atomic_fetch_add(&a.tail_cur[16 * tid], tl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b + tl <= a.tail_rows) tb = b;
                else if (b < a.tail_rows) dead = a.tail_rows - b;      // (the reservation fell off the end: what lies inside is nobody's)
            }
            SH.tail[tid] = tb; SH.tail[2 + tid] = dead;
            if (a.tail_base) a.tail_base[2 * s + tid] = tb;
        }
    }
    if (tid == 0) {                                      // requested here, needed after the sampling arithmetic
        if constexpr (ASYNC) {
            oi = s * a.row_stride; on = s * a.row_stride;
        }
        else {
            oi = atomicAdd(&a.counters[0], LRG_PAD_ROWS(rin));
            on = atomicAdd(&a.counters[1], LRG_PAD_ROWS(rnb));
        }
    }
    if (mine) {
        const int nn = half ? nc : ne, kk = half ? Ni : Nn;      // (here the first half samples the neighbours, the second the inliers)
        // only the distinct rows are gathered: a padded set's are its members in order (:240,:252), a full set's the prefix of the
        // permutation (:238,:250); the positions of the copies are the next update's business
        if (j < min(nn, kk)) {
            const int pos = nn < kk ? j : (int)lrg_sample_position((uint32_t)j, (uint32_t)nn, (uint32_t)kk, half ? LRG_PURPOSE_INLIER : LRG_PURPOSE_NEIGHBOR,
                                                                   (uint32_t)cur_seed, (uint32_t)cur_restart, (uint32_t)cur_step, k0, k1);
            // (a fresh seed: the lists were stored a moment ago by other threads -- the seed and its neighbours are still at hand in a register / in LDS)
            sh_src[half ? 0 : 1][j] = lists_ready ? (half ? cur_seed : sh_list[pos]) : lds_lists ? (half ? lds_cur : lds_cand)[pos] : (half ? cur_idx : cand_idx)[pos];
        }
    }
    if (tid == 0) {
        sh_off[0] = oi; sh_off[1] = on;
        a.slot_rows[4 * s + 0] = rin; a.slot_rows[4 * s + 1] = rnb; a.slot_rows[4 * s + 2] = oi; a.slot_rows[4 * s + 3] = on;
        big[2 * s] = is_big ? 1 : 0;
        if (is_big) big[2 * s + 1] += 1;                 // tag of this iteration's centres (LrgFusedMedians: medians in the branch launch)
    }
    if (tid < 16) sh_c[tid] = 0.f;
    LRG_LDS_BARRIER();               // (the gather reads the source indices and the row offsets from LDS)
    TRACE2(s, 4); phase(4);
    if (is_big) {
        // the nine medians of such a region come from lrg_front_big_kernel (one workgroup per channel) or from the median
        // workgroups of this launch; nothing here waits for them: the rows go out uncentred
        if (tid < 16 && !(ASYNC && a.own_medians)) a.center[s * 16 + tid] = 0.f;
        lrg_front_gather<PAD, ASYNC>(cur_target, points, obj, s, F, Ni, Nn, a, sh_src, rin, rnb, sh_off[0], sh_off[1], 0, LRG_FRONT_THREADS, ASYNC ? SH.tail[0] : -1, ASYNC ? SH.tail[1] : -1);
        TRACE2(s, 5); phase(5);
        if constexpr (ASYNC) if (a.own_medians) {   // the region's medians by this workgroup (no launch of the (slot, channel) medians)
            __syncthreads();
            lrg_front_all_medians(R, cur_idx, nc, F, reinterpret_cast<int *>(sh_flags), sh_c);
            __syncthreads();
            if constexpr (ASYNC) { if (tid < 4) lrg_st_coh4(a.center + s * 16, (unsigned)tid * 16u, *reinterpret_cast<const float4 *>(sh_c + 4 * tid)); }      // (16-byte write-through stores)
            else if (tid < 16) a.center[s * 16 + tid] = sh_c[tid];
        }
        phase(6);
        if (a.phase_ticks && tid == 0) a.phase_ticks[2 * s + 1] += wall_clock64() - tick1;
        TRACE2(s, 6); TRACE2(s, 7);
#if LRG_TRACE
        if (tid == 0 && g_lrg_trace2) { g_lrg_trace2[(long)s * 16 + 8] = nc; g_lrg_trace2[(long)s * 16 + 14] = lrg_is_stop(entry_status) || entry_status == LRG_WAIT || (entry_status == LRG_ACTIVE && S->step == 0); }
#endif
        return (rin << 16) | rnb;
    }
    const long long t_small = (ASYNC && LRG_ASYNC_DEBUG && a.phase_dbg) ? wall_clock64() : 0;
    if (wave < 9) {                                              // one wavefront per centred channel, keys in registers
        const int ch = lrg_centred_channel(wave, F);
        if (ch >= 0) {
            const LrgChanSrc cs = lrg_chan_src(R, wave, ch, F);
            const int32_t *mlist = lds_lists ? lds_cur : cur_idx;                            // (the list's copy in LDS where there is one)
            const float m = lists_ready ? cs.base[(long)cur_seed * cs.stride]                  // (a fresh seed: the median of one point is that point, :241)
                          : nc <= 256 ? lrg_median_wave_r<4>(cs.base, mlist, cs.stride, nc)
                          : nc <= 1024 ? lrg_median_wave_r<16>(cs.base, mlist, cs.stride, nc)
                                       : lrg_median_wave_r64(cs.base, cur_idx, cs.stride, nc);
            if (lane == 0) sh_c[ch] = m;
        }
        if constexpr (ASYNC && LRG_ASYNC_DEBUG) if (a.phase_dbg && tid == 0) atomicAdd(&a.phase_dbg[0], (unsigned long long)(wall_clock64() - t_small));      // (one median)
    } else {
        lrg_front_gather<PAD, ASYNC>(cur_target, points, obj, s, F, Ni, Nn, a, sh_src, rin, rnb, sh_off[0], sh_off[1], 9 * 64, LRG_FRONT_THREADS - 9 * 64, ASYNC ? SH.tail[0] : -1, ASYNC ? SH.tail[1] : -1);
        if constexpr (ASYNC && LRG_ASYNC_DEBUG) if (a.phase_dbg && tid == 9 * 64) atomicAdd(&a.phase_dbg[7], (unsigned long long)(wall_clock64() - t_small));  // (the gather)
    }
    __syncthreads();
    TRACE2(s, 5); phase(5);
    // the branch kernels and the next update centre with it (:243-247,:271,:275)
#ifdef LRG_SPEC_TRACE
    if (spec && tid == 0) LRG_SPEC_EV(a, 7, s, cur_seed, ((long long)sh_src[0][0] << 32) | (unsigned)sh_src[1][0], ((long long)__float_as_uint(sh_c[0]) << 32) | __float_as_uint(sh_c[1]));
#endif
    if constexpr (ASYNC) { if (tid < 4) lrg_st_coh4(a.center + s * 16, (unsigned)tid * 16u, *reinterpret_cast<const float4 *>(sh_c + 4 * tid)); }
    else if (tid < 16) a.center[s * 16 + tid] = sh_c[tid];
    TRACE2(s, 6);
    if (a.phase_ticks && tid == 0) a.phase_ticks[2 * s + 1] += wall_clock64() - tick1;
    TRACE2(s, 7);
#if LRG_TRACE
    if (tid == 0 && g_lrg_trace2) { g_lrg_trace2[(long)s * 16 + 8] = nc; g_lrg_trace2[(long)s * 16 + 14] = lrg_is_stop(entry_status) || entry_status == LRG_WAIT || (entry_status == LRG_ACTIVE && S->step == 0); }
#endif
    return (rin << 16) | rnb;
}

// The greedy front as a launch of its own (lrg_grow_step_packed): one workgroup per slot.
__global__ __launch_bounds__(LRG_FRONT_THREADS) void lrg_front_greedy_kernel(LrgSlot *slots, LrgRoom *rooms, int n_slots, LrgGrowParams prm,
                                                                             LrgFrontArgs a, int32_t *big) {
    __shared__ LrgFrontShared SH;
#if LRG_FRONT_EXCLUSIVE
    asm volatile("" ::: "v127");         // the kernel is accounted 128 VGPRs: four of its waves fill a SIMD's register file, a CU to itself
#endif
    if (threadIdx.x == 0) SH.upd[0] = 0;
    __syncthreads();
    (void)lrg_front_greedy_slot<false>(SH, slots, rooms, n_slots, prm, a, big, (int)blockIdx.x);
}

// Medians of regions above LRG_FRONT_SMALL points: one workgroup per (slot, centred channel), keys in registers (two
// global round trips), two bits per bisection step.  Nothing but the centre is written: the rows were gathered uncentred.
__global__ __launch_bounds__(1024) void lrg_front_big_kernel(const LrgSlot *slots, const LrgRoom *rooms, LrgGrowParams prm,
                                                              LrgFrontArgs a, int32_t *big) {
    __shared__ __attribute__((aligned(16))) int sh[LRG_SAMPLED_LDS_INTS(1024)];      // (>= LRG_RADIX_LDS_INTS(1))
#if LRG_BIG_EXCLUSIVE
    asm volatile("" ::: "v127");
#endif
    const int s = blockIdx.x, tid = threadIdx.x;
    if (big[2 * s] == 0) return;
    const long long tickb = a.phase_ticks ? wall_clock64() : 0;
    const LrgSlot *S = &slots[s];
    const LrgRoom *R = &rooms[S->room];
    const int F = prm.feature_size;
    const int nc = S->nc;
    const int ch = lrg_centred_channel(blockIdx.y, F);
    if (ch < 0) return;
    const LrgChanSrc cs = lrg_chan_src(R, blockIdx.y, ch, F);
    if (nc <= 256) {                                 // one wavefront, keys in registers, no barrier
        if (tid >= 64) return;
        const float mw = lrg_median_wave_r<4>(cs.base, S->cur_idx, cs.stride, nc);
        if (tid == 0) {
            a.center[s * 16 + ch] = mw;
            if (a.phase_ticks && blockIdx.y == 0) a.phase_ticks[2 * s + 1] += wall_clock64() - tickb;
        }
        return;
    }
    float m;
    const int chs[1] = {0};
    float mm[1];
    if (nc <= 4096) { lrg_median_block_radix<4, 1024, 1>(cs.base, chs, S->cur_idx, cs.stride, nc, sh, mm); m = mm[0]; }
    else if (nc <= 16 * 1024) { lrg_median_block_radix<16, 1024, 1>(cs.base, chs, S->cur_idx, cs.stride, nc, sh, mm); m = mm[0]; }
    else if (nc <= LRG_MED_REGS) {
#if LRG_MED48_BISECT == 2
        // 48 keys per thread: a radix pass is 48 LDS atomics per thread (49 k per workgroup), a bisection step 144 compares per
        // thread -- two sampled pivots first, then the selection among the eighth of the keys between them
        m = lrg_median_block_sampled<48>(cs.base, S->cur_idx, cs.stride, nc, sh);
#elif LRG_MED48_BISECT
        if (tid < 64) sh[tid] = tid == 0 ? -1 : 0;
        __syncthreads();
        m = lrg_median_block_regs<48>(cs.base, S->cur_idx, cs.stride, nc, sh);
#else
        lrg_median_block_radix<48, 1024, 1>(cs.base, chs, S->cur_idx, cs.stride, nc, sh, mm); m = mm[0];
#endif
    }
    else {
        if (tid < 64) sh[tid] = tid == 0 ? -1 : 0;
        __syncthreads();
        const int k2 = nc >> 1, k1r = (nc & 1) ? k2 : k2 - 1;
        uint32_t ka, kb;
        lrg_select2(nullptr, false, cs.base, S->cur_idx, cs.stride, 0, nc, k1r, k2, sh, &ka, &kb);
        const float lo = lrg_key2f(ka), hi = lrg_key2f(kb);
        m = (nc & 1) ? hi : __fmul_rn(__fadd_rn(lo, hi), 0.5f);
    }
    if (tid == 0) {
        a.center[s * 16 + ch] = m;
        if (a.phase_ticks && blockIdx.y == 0) a.phase_ticks[2 * s + 1] += wall_clock64() - tickb;
    }
}
