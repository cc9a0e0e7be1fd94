// Region-grow loop kernels (gfx950): the per-step bookkeeping around the LrgNet forward.
//
// Reference: test_region_grow.py:175-316 (greedy) and test_random_restart.py:141-303 (restarts).
// The reference runs ONE region of ONE room per step on the host; here S "slots" (region instances)
// advance in lock-step, one workgroup per slot (or per slot group), all state device-resident.
#include "lrg_common.h"
#include "lrg_rng.h"
#include "lrg_median.h"
#include "lrg_fused.h"
#include "lrg_fused_tile.inl"

#define LRG_SCAN_THREADS 1024

#ifndef LRG_TRACE
#define LRG_TRACE 0
#endif
#if LRG_TRACE
__device__ long long *g_lrg_trace2 = nullptr;
extern "C" void lrg_set_trace2(long long *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lrg_trace2), &p, sizeof(p)); }
#define TRACE2(slot, i) do { if (threadIdx.x == 0 && g_lrg_trace2) g_lrg_trace2[(long)(slot) * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TRACE2(slot, i)
#endif

__device__ __forceinline__ bool lrg_is_stop(int st) { return st >= LRG_STOP_NONEIGHBOR && st <= LRG_STOP_MAXSTEPS; }

// ------------------------------------------------------------------------------------------------
// voxelise + voxel hash
// ------------------------------------------------------------------------------------------------
__global__ void lrg_voxelize_kernel(const float *points, int n, int F, float res, int32_t *vox) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vox[3 * i + 0] = lrg_voxel_of(points[(long)i * F + 0], res);
    vox[3 * i + 1] = lrg_voxel_of(points[(long)i * F + 1], res);
    vox[3 * i + 2] = lrg_voxel_of(points[(long)i * F + 2], res);
}

__global__ void lrg_voxel_pack_kernel(const int32_t *vox, int n, int ox, int oy, int oz, uint32_t *pvox, int32_t *overflow) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = vox[3 * i] - ox, y = vox[3 * i + 1] - oy, z = vox[3 * i + 2] - oz;
    if ((unsigned)x > 2047u || (unsigned)y > 2047u || (unsigned)z > 1023u) atomicOr(overflow, 1);
    pvox[i] = (uint32_t)min(max(x, 0), 2047) | ((uint32_t)min(max(y, 0), 2047) << 11) | ((uint32_t)min(max(z, 0), 1023) << 22);
}

__global__ void lrg_voxel_grid_kernel(const int32_t *vox, int n, int ox, int oy, int oz, int gx, int gy, int gz, int32_t *grid) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = vox[3 * i] - ox, y = vox[3 * i + 1] - oy, z = vox[3 * i + 2] - oz;
    if ((unsigned)x < (unsigned)gx && (unsigned)y < (unsigned)gy && (unsigned)z < (unsigned)gz) grid[((long)z * gy + y) * gx + x] = i;
}

__global__ void lrg_hash_build_kernel(const int32_t *vox, int n, uint64_t *keys, int32_t *vals, int mask, int32_t *dup) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t key = lrg_pack_voxel(vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]);
    if (key == LRG_HASH_EMPTY) { atomicOr(dup, 2); return; }
    unsigned h = (unsigned)lrg_fmix64(key) & (unsigned)mask;
    for (int probe = 0; probe <= mask; ++probe) {
        unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long *>(&keys[h]),
                                            (unsigned long long)LRG_HASH_EMPTY, (unsigned long long)key);
        if (prev == LRG_HASH_EMPTY) { vals[h] = i; return; }
        if (prev == key) { atomicOr(dup, 1); return; }
        h = (h + 1) & (unsigned)mask;
    }
    atomicOr(dup, 4);
}

// ------------------------------------------------------------------------------------------------
// (re)binding a slot group to a room on the device: what a host-side struct upload would do, without the upload (a
// pageable host-to-device copy blocks the host until the stream has drained, which starves the other lanes)
// ------------------------------------------------------------------------------------------------
// (a device function: the free-running kernel rebinds a slot whose room is finished to the next room of its queue itself)
__device__ __noinline__ void lrg_bind_group_device(LrgSlot *slots, LrgRoom *rooms, int first_slot, int group_size, int room, int reset_room,
                                                      int clear_masks) {
    const int tid = threadIdx.x;
    if (room >= 0 && reset_room) {                                            // test_region_grow.py:176-178 for a fresh pass
        LrgRoom *R = &rooms[room];
        const int n = R->n;
        for (int i = tid; i < n; i += blockDim.x) { R->visited[i] = 0; R->label[i] = 0; }
        if (tid == 0) { R->next_cluster_id = 1; R->seed_cursor = 0; R->n_regions = 0; R->done = 0; }
    }
    for (int s = first_slot; s < first_slot + group_size; ++s) {
        LrgSlot *S = &slots[s];
        if (clear_masks && S->room >= 0) {                                    // the room the slot is leaving bounds what can be set
            const int n = rooms[S->room].n;
            for (int i = tid; i < n; i += blockDim.x) S->cur[i] = 0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        for (int s = first_slot; s < first_slot + group_size; ++s) {
            LrgSlot *S = &slots[s];
            S->room = room;
            S->status = room >= 0 ? LRG_WAIT : LRG_IDLE;                      // lrg_advance / the front kernel pick the first seed
            S->seed = -1;
            S->restart = 0; S->step = 0; S->steps_total = 0; S->stuck = 0;
            S->updated = -1; S->count = -1; S->best_count = -1; S->pad = 0;
            S->acc_add = -1; S->acc_rmv = -1; S->ml_score = 0.0; S->ml_best = 0.0;
            S->scan_cnt = 0; S->query = 0;
            for (int d = 0; d < 3; ++d) { S->scan_mn[d] = INT_MAX; S->scan_mx[d] = INT_MIN; }
            S->spec_pos = INT_MAX; S->spec_flags = 0;
        }
    // (the free-running kernel serves these slots right behind this barrier, every wavefront reading the fields thread 0 has just stored: __syncthreads()
    //  waits for LDS traffic only -- `s_waitcnt lgkmcnt(0); s_barrier` -- so the stores are waited for explicitly)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__global__ __launch_bounds__(1024) void lrg_bind_group_kernel(LrgSlot *slots, LrgRoom *rooms, int first_slot, int group_size, int room,
                                                              int reset_room, int clear_masks) {
    lrg_bind_group_device(slots, rooms, first_slot, group_size, room, reset_room, clear_masks);
}

// ------------------------------------------------------------------------------------------------
// block helpers (1024 threads = 16 waves)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lrg_block_sum(int v, int *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lrg_lane() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int nw = blockDim.x >> 6;
    int r = 0;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

// count + signed min / max of three coordinates over the block in ONE barrier: DPP reductions inside each wavefront, the
// 16 x 7 partials through LDS, combined by thread 0 (the only consumer).  The seven separate lrg_block_* calls this replaces
// cost ~2 k cycles each (six LDS-crossbar shuffles, two barriers, a serial read of the partials).
__device__ __forceinline__ void lrg_block_bbox(int &cnt, int &mn0, int &mn1, int &mn2, int &mx0, int &mx1, int &mx2, int *red7) {
    const unsigned B = 0x80000000u;                     // order-preserving signed -> unsigned
    cnt = lrg_wave_sum_i32(cnt);
    mn0 = (int)(lrg_wave_min_u32((unsigned)mn0 ^ B) ^ B); mn1 = (int)(lrg_wave_min_u32((unsigned)mn1 ^ B) ^ B);
    mn2 = (int)(lrg_wave_min_u32((unsigned)mn2 ^ B) ^ B);
    mx0 = (int)(lrg_wave_max_u32((unsigned)mx0 ^ B) ^ B); mx1 = (int)(lrg_wave_max_u32((unsigned)mx1 ^ B) ^ B);
    mx2 = (int)(lrg_wave_max_u32((unsigned)mx2 ^ B) ^ B);
    const int w = threadIdx.x >> 6;
    if (lrg_lane() == 0) {
        int *r = red7 + 8 * w;
        r[0] = cnt; r[1] = mn0; r[2] = mn1; r[3] = mn2; r[4] = mx0; r[5] = mx1; r[6] = mx2;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // the (at most 16) wavefronts' partials: one per lane, combined by four shuffle steps (thread 0 reading them one after the other
        // was ~1 000 cycles of the mask update's 1.3 us reduction, profiles/r03_update_subphases.txt)
        const int nw = blockDim.x >> 6, lane = threadIdx.x;
        const int *r = red7 + 8 * min(lane, nw - 1);
        const bool on = lane < nw;
        cnt = on ? r[0] : 0;
        mn0 = on ? r[1] : INT_MAX; mn1 = on ? r[2] : INT_MAX; mn2 = on ? r[3] : INT_MAX;
        mx0 = on ? r[4] : INT_MIN; mx1 = on ? r[5] : INT_MIN; mx2 = on ? r[6] : INT_MIN;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            cnt += __shfl_xor(cnt, o);
            mn0 = min(mn0, __shfl_xor(mn0, o)); mn1 = min(mn1, __shfl_xor(mn1, o)); mn2 = min(mn2, __shfl_xor(mn2, o));
            mx0 = max(mx0, __shfl_xor(mx0, o)); mx1 = max(mx1, __shfl_xor(mx1, o)); mx2 = max(mx2, __shfl_xor(mx2, o));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// scan of the updated mask (test_region_grow.py:292-293): one workgroup per (slot, 4096-point chunk)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LRG_SCAN_THREADS) void lrg_bbox_scan_kernel(LrgSlot *slots, const LrgRoom *rooms) {
    __shared__ int red[16 * 8];
    LrgSlot *S = &slots[blockIdx.x];
    if (S->status != LRG_ACTIVE || S->updated < 0 || S->room < 0) return;
    const LrgRoom *R = &rooms[S->room];
    const int n = R->n;
    const int i0 = blockIdx.y * LRG_SCAN_CHUNK;
    if (i0 >= n) return;
    const uint8_t *cur = S->cur;
    const int32_t *vox = R->voxels;
    int cnt = 0;
    int mn0 = INT_MAX, mn1 = INT_MAX, mn2 = INT_MAX, mx0 = INT_MIN, mx1 = INT_MIN, mx2 = INT_MIN;
    {
        const int ib = i0 + 4 * threadIdx.x;
        int m[4], a[4], b[4], c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                 // unconditional loads (clamped): no dependent round trips
            const int i = min(ib + k, n - 1);
            m[k] = cur[i]; a[k] = vox[3 * i]; b[k] = vox[3 * i + 1]; c[k] = vox[3 * i + 2];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (ib + k < n && m[k]) {
                ++cnt;
                mn0 = min(mn0, a[k]); mn1 = min(mn1, b[k]); mn2 = min(mn2, c[k]);
                mx0 = max(mx0, a[k]); mx1 = max(mx1, b[k]); mx2 = max(mx2, c[k]);
            }
        }
    }
    lrg_block_bbox(cnt, mn0, mn1, mn2, mx0, mx1, mx2, red);
    if (threadIdx.x != 0 || cnt == 0) return;
    atomicAdd(&S->scan_cnt, cnt);
    atomicMin(&S->scan_mn[0], mn0); atomicMin(&S->scan_mn[1], mn1); atomicMin(&S->scan_mn[2], mn2);
    atomicMax(&S->scan_mx[0], mx0); atomicMax(&S->scan_mx[1], mx1); atomicMax(&S->scan_mx[2], mx2);
}

// The same scan by ONE workgroup over the whole room (fused into lrg_advance for greedy growing: a room is at most a few
// tens of thousands of points, and a launch of its own costs more than the 0.6 MB this reads).  Per-thread partials over
// all chunks, then a single round of block reductions; thread 0 stores the result where lrg_stop_logic expects it.
__device__ void lrg_scan_mask_block(LrgSlot *S, const LrgRoom *R, int *red) {
    const int n = R->n;
    const uint8_t *cur = S->cur;
    const int32_t *vox = R->voxels;
    int cnt = 0;
    int mn0 = INT_MAX, mn1 = INT_MAX, mn2 = INT_MAX, mx0 = INT_MIN, mx1 = INT_MIN, mx2 = INT_MIN;
    // 16 points per thread per trip, every load of a trip issued before the first use (the trip count is a run-time value:
    // left to itself the loop pays one memory round trip per 4 points)
    for (int i0 = 0; i0 < n; i0 += 16 * (int)blockDim.x) {
        int m[16], a[16], b[16], c[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {                // unconditional loads (clamped): no dependent round trips
            const int i = min(i0 + (k >> 2) * 4 * (int)blockDim.x + 4 * (int)threadIdx.x + (k & 3), n - 1);
            m[k] = cur[i]; a[k] = vox[3 * i]; b[k] = vox[3 * i + 1]; c[k] = vox[3 * i + 2];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = i0 + (k >> 2) * 4 * (int)blockDim.x + 4 * (int)threadIdx.x + (k & 3);
            if (i < n && m[k]) {
                ++cnt;
                mn0 = min(mn0, a[k]); mn1 = min(mn1, b[k]); mn2 = min(mn2, c[k]);
                mx0 = max(mx0, a[k]); mx1 = max(mx1, b[k]); mx2 = max(mx2, c[k]);
            }
        }
    }
    lrg_block_bbox(cnt, mn0, mn1, mn2, mx0, mx1, mx2, red);
    if (threadIdx.x == 0) {
        S->scan_cnt = cnt;
        S->scan_mn[0] = mn0; S->scan_mn[1] = mn1; S->scan_mn[2] = mn2;
        S->scan_mx[0] = mx0; S->scan_mx[1] = mx1; S->scan_mx[2] = mx2;
    }
}

// stop / bbox decision of the step just taken (:291-306), from the scan results; run by one thread of lrg_advance
__device__ void lrg_stop_logic(LrgSlot *S) {
    const int updated = S->updated;
    const int cnt = S->scan_cnt;
    const int mn0 = S->scan_mn[0], mn1 = S->scan_mn[1], mn2 = S->scan_mn[2];
    const int mx0 = S->scan_mx[0], mx1 = S->scan_mx[1], mx2 = S->scan_mx[2];
    S->scan_cnt = 0;
    S->scan_mn[0] = S->scan_mn[1] = S->scan_mn[2] = INT_MAX;
    S->scan_mx[0] = S->scan_mx[1] = S->scan_mx[2] = INT_MIN;
    S->updated = -1;
    S->count = cnt;
    if (!updated) { S->status = LRG_STOP_NOEXPAND; S->last_reason = LRG_STOP_NOEXPAND; return; }   // :304-306
    if (cnt == 0) { S->status = LRG_STOP_EMPTY; S->last_reason = LRG_STOP_EMPTY; return; }          // :292 would raise
    S->mn[0] = mn0; S->mn[1] = mn1; S->mn[2] = mn2;                                                  // :292-293
    S->mx[0] = mx0; S->mx[1] = mx1; S->mx[2] = mx2;
    bool grew = mn0 < S->seq_mn[0] || mn1 < S->seq_mn[1] || mn2 < S->seq_mn[2] ||
                mx0 > S->seq_mx[0] || mx1 > S->seq_mx[1] || mx2 > S->seq_mx[2];                     // :294
    if (!grew) {
        if (S->stuck >= 1) { S->status = LRG_STOP_STUCK; S->last_reason = LRG_STOP_STUCK; return; }  // :295-297
        S->stuck += 1;                                                                               // :299
    } else {
        S->stuck = 0;                                                                                // :301
    }
    S->seq_mn[0] = min(S->seq_mn[0], mn0); S->seq_mn[1] = min(S->seq_mn[1], mn1); S->seq_mn[2] = min(S->seq_mn[2], mn2);
    S->seq_mx[0] = max(S->seq_mx[0], mx0); S->seq_mx[1] = max(S->seq_mx[1], mx1); S->seq_mx[2] = max(S->seq_mx[2], mx2);
}

// ------------------------------------------------------------------------------------------------
// commit / next seed / reset     (test_region_grow.py:186-217, test_random_restart.py:169-197)
// One workgroup per group of `group_size` slots sharing a room and a seed.
// ------------------------------------------------------------------------------------------------
__device__ void lrg_reset_slot(LrgSlot *S, const LrgRoom *R, int seed, int restart) {
    // cooperative: clear the mask, set the seed (:197-204)
    for (int i = threadIdx.x; i < R->n; i += blockDim.x) S->cur[i] = (i == seed) ? 1 : 0;
    if (threadIdx.x == 0) {
        S->seed = seed;
        S->restart = restart;
        S->step = 0;
        S->stuck = 0;
        S->updated = -1;
        S->nc = 1; S->ne = 0; S->count = 1;
        for (int d = 0; d < 3; ++d) {
            int v = R->voxels[3 * seed + d];
            S->mn[d] = v; S->mx[d] = v; S->seq_mn[d] = v; S->seq_mx[d] = v;
        }
        S->target = R->obj_id ? R->obj_id[seed] : 0;
        S->acc_add = -1; S->acc_rmv = -1;
        S->ml_score = 0.0;
        S->pad = 0;
        S->scan_cnt = 0;
        S->scan_mn[0] = S->scan_mn[1] = S->scan_mn[2] = INT_MAX;
        S->scan_mx[0] = S->scan_mx[1] = S->scan_mx[2] = INT_MIN;
        S->status = LRG_ACTIVE;
    }
}

#define LRG_SEED_TRIES 64   // isolated seeds committed per call before the search is resumed by the next one
// The body of lrg_advance for the group whose first slot is g0, run by one 1024-thread workgroup (every return is
// workgroup-uniform).  FUSE_SCAN: do lrg_bbox_stop's scan here (one workgroup per slot; lrg_grow_step, greedy growing).
template <bool FUSE_SCAN>
__device__ void lrg_advance_group(LrgSlot *slots, LrgRoom *rooms, int n_slots, const LrgGrowParams &prm, int64_t *stats, int g0) {
    __shared__ int sh_next;
    __shared__ int sh_flag;
    __shared__ int sh_list[32];
    const int G = prm.group_size, RST = prm.restarts;
    if (g0 >= n_slots) return;
    LrgSlot *S0 = &slots[g0];
    if (S0->room < 0) return;
    LrgRoom *R = &rooms[S0->room];
    const int n = R->n;
    if (S0->status == LRG_DONE || S0->status == LRG_IDLE) return;

    // ---- phase 0: stop / stuck decision of the step just taken, from lrg_bbox_stop's scan ----
    TRACE2(g0, 9);
    if (FUSE_SCAN) {
        __shared__ int red[16 * 8];
        for (int s = 0; s < G && g0 + s < n_slots; ++s) {
            LrgSlot *S = &slots[g0 + s];
            if (S->status == LRG_ACTIVE && S->updated >= 0) lrg_scan_mask_block(S, R, red);     // (uniform across the block)
        }
        __syncthreads();
    }
    TRACE2(g0, 10);
    if (threadIdx.x == 0) {
        int work = 0;
        for (int s = 0; s < G && g0 + s < n_slots; ++s) {
            LrgSlot *S = &slots[g0 + s];
            if (S->status == LRG_ACTIVE && S->updated >= 0) lrg_stop_logic(S);
            if (S->status != LRG_ACTIVE) work = 1;
        }
        sh_flag = work;
    }
    __syncthreads();
    if (!sh_flag) return;      // every slot of the group keeps growing: nothing to bank, commit or reseed
    __syncthreads();

    TRACE2(g0, 11);
    // ---- phase 1: bank finished grows, start the slot's next restart (restart :173-175,:187-197) ----
    for (int s = 0; s < G && g0 + s < n_slots; ++s) {
        LrgSlot *S = &slots[g0 + s];
        int st = S->status;   // uniform across the block
        if (!lrg_is_stop(st)) continue;
        if (RST > 1) {
            bool better = S->best_count < 0 || (prm.scoring == 1 ? S->ml_score > S->ml_best : S->count > S->best_count);   // first max wins (numpy.argmax, :177)
            if (better) {
                for (int i = threadIdx.x; i < n; i += blockDim.x) S->best[i] = S->cur[i];
            }
            __syncthreads();
            if (threadIdx.x == 0 && better) { S->best_count = S->count; S->best_restart = S->restart; S->ml_best = S->ml_score; }
        }
        int next_restart = S->restart + G;
        __syncthreads();
        if (RST > 1 && next_restart < RST) {
            lrg_reset_slot(S, R, S->seed, next_restart);
        } else if (threadIdx.x == 0) {
            S->status = LRG_WAIT;
        }
        __syncthreads();
    }

    TRACE2(g0, 12);
    // ---- phase 2: when every slot of the group waits, commit the seed ----
    if (threadIdx.x == 0) {
        int all_wait = 1;
        for (int s = 0; s < G && g0 + s < n_slots; ++s)
            if (slots[g0 + s].status != LRG_WAIT) all_wait = 0;
        sh_flag = all_wait;
    }
    __syncthreads();
    if (!sh_flag) return;

    if (S0->seed >= 0) {
        // winner: max score, earliest restart ordinal (restart :177); greedy: the slot's own mask
        int win = 0, wcount = -1, wrest = INT_MAX, steps = 0;
        double wscore = 0.0;
        const bool ml = prm.scoring == 1 && RST > 1;
        for (int s = 0; s < G && g0 + s < n_slots; ++s) {
            const LrgSlot *S = &slots[g0 + s];
            steps += S->steps_total;
            int c = RST > 1 ? S->best_count : S->count;
            int r = RST > 1 ? S->best_restart : S->restart;
            if (c < 0) continue;
            const bool first = wcount < 0;
            const bool wins = ml ? (first || S->ml_best > wscore || (S->ml_best == wscore && r < wrest))
                                 : (c > wcount || (c == wcount && r < wrest));
            if (wins) { win = s; wcount = c; wrest = r; wscore = S->ml_best; }
        }
        const LrgSlot *W = &slots[g0 + win];
        const uint8_t *mask = RST > 1 ? W->best : W->cur;
        const int labeled = wcount > prm.cluster_threshold;                       // :213
        const int cid = R->next_cluster_id;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            if (mask[i]) {
                R->visited[i] = 1;                                                // :212
                if (labeled) R->label[i] = cid;                                   // :214
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            // reason printed by the reference is that of the call completing the seed: restart ordinal RST-1
            int last_slot = (RST - 1) % G;
            if (g0 + last_slot >= n_slots) last_slot = 0;
            int32_t *log = R->region_log + LRG_LOG_WORDS * (long)R->n_regions;
            log[0] = S0->seed; log[1] = steps; log[2] = wcount; log[3] = slots[g0 + last_slot].last_reason;
            log[4] = labeled; log[5] = wrest;
            log[6] = slots[g0 + last_slot].acc_add; log[7] = slots[g0 + last_slot].acc_rmv;
            R->n_regions += 1;
            if (labeled) R->next_cluster_id = cid + 1;                            // :215
            if (stats) atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), 1ULL);
        }
        __syncthreads();
    }

    TRACE2(g0, 13);
    // ---- next unvisited seed in curvature order (:186-188) ----
    // The first box query of a fresh seed (:221-229 with mn = mx = the seed's voxel) is answered here from the room's voxel
    // hash: an equalised room has one point per voxel, so the candidates are the unvisited points of the 26 surrounding
    // voxels, in index order.  A seed without any (:233-235, 'noneighbor') is committed on the spot as a one-point region
    // and the search goes on, so a slot never spends an iteration on it.
    int cursor = R->seed_cursor;
    int seed = -1, n_cand = 0;
    for (int tries = 0; tries < LRG_SEED_TRIES; ++tries) {
        int found = -1;
        while (cursor < n) {
            int pos = cursor + threadIdx.x;
            int cand = INT_MAX;
            if (pos < n && !R->visited[R->order[pos]]) cand = pos;
            if (threadIdx.x == 0) sh_next = INT_MAX;
            __syncthreads();
            if (cand != INT_MAX) atomicMin(&sh_next, cand);
            __syncthreads();
            int best = sh_next;
            __syncthreads();
            if (best != INT_MAX) { found = best; break; }
            cursor += blockDim.x;
        }
        if (found < 0) {
            if (threadIdx.x == 0) {
                R->seed_cursor = n;
                R->done = 1;
                for (int s = 0; s < G && g0 + s < n_slots; ++s) slots[g0 + s].status = LRG_DONE;
                if (stats) {
                    unsigned long long k = atomicAdd(reinterpret_cast<unsigned long long *>(&stats[1]), 1ULL);
                    stats[4 + (k % LRG_DONE_RING)] = g0;
                }
            }
            return;
        }
        const int sd = R->order[found];
        cursor = found + 1;
        if (!R->hash_keys) { seed = sd; n_cand = -1; break; }            // no hash: leave the query to lrg_box_query
        // 26 probes by the first lanes of wave 0, then an ordered (ascending index) list in LDS
        if (threadIdx.x < 64) {
            int idx = -1;
            if (threadIdx.x < 27 && threadIdx.x != 13) {
                const int dx = threadIdx.x / 9 - 1, dy = (threadIdx.x / 3) % 3 - 1, dz = threadIdx.x % 3 - 1;
                const int32_t *v = R->voxels + 3 * (long)sd;
                idx = lrg_hash_lookup(R->hash_keys, R->hash_vals, R->hash_mask, lrg_pack_voxel(v[0] + dx, v[1] + dy, v[2] + dz));
                if (idx >= 0 && (R->visited[idx] || idx == sd)) idx = -1;
            }
            int rank = 0, total = 0;
            for (int l = 0; l < 27; ++l) {
                const int o = __shfl(idx, l);
                if (o >= 0) { ++total; if (idx >= 0 && o < idx) ++rank; }
            }
            if (idx >= 0) sh_list[rank] = idx;
            if (threadIdx.x == 0) sh_flag = total;
        }
        __syncthreads();
        const int total = sh_flag;
        __syncthreads();
        if (total > 0) { seed = sd; n_cand = total; break; }
        // no neighbour: the region is the seed alone (:233-235 -> :210-217)
        if (threadIdx.x == 0) {
            const int labeled = 1 > prm.cluster_threshold;
            R->visited[sd] = 1;
            if (labeled) { R->label[sd] = R->next_cluster_id; R->next_cluster_id += 1; }
            int32_t *log = R->region_log + LRG_LOG_WORDS * (long)R->n_regions;
            log[0] = sd; log[1] = 0; log[2] = 1; log[3] = LRG_STOP_NONEIGHBOR; log[4] = labeled; log[5] = 0; log[6] = -1; log[7] = -1;
            R->n_regions += 1;
            if (stats) atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), 1ULL);
        }
        __syncthreads();
    }
    TRACE2(g0, 14);
    if (threadIdx.x == 0) R->seed_cursor = cursor;
    if (seed < 0) {
        // try budget spent on isolated points: park the group as a fresh binding, the next call continues the search
        if (threadIdx.x == 0)
            for (int s = 0; s < G && g0 + s < n_slots; ++s) { slots[g0 + s].seed = -1; slots[g0 + s].status = LRG_WAIT; slots[g0 + s].count = -1; }
        return;
    }
    __syncthreads();
    for (int s = 0; s < G && g0 + s < n_slots; ++s) {
        LrgSlot *S = &slots[g0 + s];
        if (threadIdx.x == 0) { S->steps_total = 0; S->best_count = -1; S->best_restart = INT_MAX; S->last_reason = 0; }
        if (s < RST) {
            lrg_reset_slot(S, R, seed, s);
            if (n_cand > 0) {                                          // the slot's lists are ready: lrg_box_query skips it
                if ((int)threadIdx.x < n_cand) S->cand_idx[threadIdx.x] = sh_list[threadIdx.x];
                __syncthreads();
                if (threadIdx.x == 0) { S->cur_idx[0] = seed; S->nc = 1; S->ne = n_cand; S->pad = 1; }
            }
        } else if (threadIdx.x == 0) {
            S->seed = seed; S->status = LRG_WAIT; S->count = -1;
        }
        __syncthreads();
    }
    TRACE2(g0, 15);
}

template <bool FUSE_SCAN>
__global__ __launch_bounds__(LRG_SCAN_THREADS) void lrg_advance_kernel(LrgSlot *slots, LrgRoom *rooms, int n_slots,
                                                                        LrgGrowParams prm, int64_t *stats) {
    lrg_advance_group<FUSE_SCAN>(slots, rooms, n_slots, prm, stats, blockIdx.x * prm.group_size);
}

// ------------------------------------------------------------------------------------------------
// dilated voxel-box query + ordered compaction   (test_region_grow.py:221-235)
// two passes over (slot, 4096-point chunk) workgroups: count, then compact at the chunk's global offset
// ------------------------------------------------------------------------------------------------
struct LrgBoxFlags { int c, e; };   // bit k: point 4*tid+k of the chunk is current / is an expand candidate

__device__ __forceinline__ LrgBoxFlags lrg_box_flags(const LrgSlot *S, const LrgRoom *R, int i0, int n) {
    const int lo0 = S->mn[0] - 1, lo1 = S->mn[1] - 1, lo2 = S->mn[2] - 1;     // :222-225
    const int hi0 = S->mx[0] + 1, hi1 = S->mx[1] + 1, hi2 = S->mx[2] + 1;
    const uint8_t *cur = S->cur, *visited = R->visited;
    const int32_t *vox = R->voxels;
    LrgBoxFlags f = {0, 0};
    const int ib = i0 + 4 * threadIdx.x;
    // all loads unconditional (clamped index): dependent predicated loads would serialise into 12 round trips
    int c[4], vis[4], a[4], b[4], d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = min(ib + k, n - 1);
        c[k] = cur[i]; vis[k] = visited[i];
        a[k] = vox[3 * i]; b[k] = vox[3 * i + 1]; d[k] = vox[3 * i + 2];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (ib + k < n) {
            if (c[k]) f.c |= 1 << k;
            else if (!vis[k] && a[k] >= lo0 && a[k] <= hi0 && b[k] >= lo1 && b[k] <= hi1 && d[k] >= lo2 && d[k] <= hi2)
                f.e |= 1 << k;                                                  // :226-228
        }
    }
    return f;
}

__global__ __launch_bounds__(LRG_SCAN_THREADS) void lrg_box_count_kernel(LrgSlot *slots, const LrgRoom *rooms) {
    __shared__ int red[16];
    LrgSlot *S = &slots[blockIdx.x];
    const bool go = S->status == LRG_ACTIVE && S->room >= 0 && S->pad != 1;
    if (blockIdx.y == 0 && threadIdx.x == 0) S->query = go ? 1 : 0;           // read by the compaction pass
    if (!go) return;
    const LrgRoom *R = &rooms[S->room];
    const int n = R->n;
    const int i0 = blockIdx.y * LRG_SCAN_CHUNK;
    if (i0 >= n) return;
    LrgBoxFlags f = lrg_box_flags(S, R, i0, n);
    int packed = __popc(f.c) | (__popc(f.e) << 16);
    packed = lrg_block_sum(packed, red);
    if (threadIdx.x == 0) { S->chunk_cnt[2 * blockIdx.y] = packed & 0xFFFF; S->chunk_cnt[2 * blockIdx.y + 1] = packed >> 16; }
}

__global__ __launch_bounds__(LRG_SCAN_THREADS) void lrg_box_compact_kernel(LrgSlot *slots, const LrgRoom *rooms,
                                                                            LrgGrowParams prm) {
    __shared__ int wtot[16];
    LrgSlot *S = &slots[blockIdx.x];
    if (S->room < 0 || S->query != 1) return;
    const LrgRoom *R = &rooms[S->room];
    const int n = R->n;
    const int i0 = blockIdx.y * LRG_SCAN_CHUNK;
    if (i0 >= n) return;
    const int nchunk = (n + LRG_SCAN_CHUNK - 1) / LRG_SCAN_CHUNK;
    int base_c = 0, base_e = 0, tot_c = 0, tot_e = 0;
    for (int c = 0; c < nchunk; ++c) {
        const int cc = S->chunk_cnt[2 * c], ce = S->chunk_cnt[2 * c + 1];
        if (c < (int)blockIdx.y) { base_c += cc; base_e += ce; }
        tot_c += cc; tot_e += ce;
    }
    LrgBoxFlags f = lrg_box_flags(S, R, i0, n);
    // exclusive scan of the packed per-thread counts over the 1024 threads
    const int lane = lrg_lane(), wave = threadIdx.x >> 6;
    const int mine = __popc(f.c) | (__popc(f.e) << 16);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wtot[w];
    const int excl = woff + incl - mine;
    int pc = base_c + (excl & 0xFFFF), pe = base_e + (excl >> 16);
    const int ib = i0 + 4 * threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (f.c >> k & 1) S->cur_idx[pc++] = ib + k;
        if (f.e >> k & 1) S->cand_idx[pe++] = ib + k;
    }
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        S->pad = 1;
        S->nc = tot_c;
        S->ne = tot_e;
        if (tot_e == 0) {                                                       // :233-235
            S->status = LRG_STOP_NONEIGHBOR; S->last_reason = LRG_STOP_NONEIGHBOR; S->count = tot_c;
        } else if (prm.max_region_steps > 0 && S->step >= prm.max_region_steps) {
            S->status = LRG_STOP_MAXSTEPS; S->last_reason = LRG_STOP_MAXSTEPS; S->count = tot_c;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per-channel median of the current points   (numpy.median, test_region_grow.py:241)
// one workgroup per (slot, channel); radix select on order-preserving keys
// ------------------------------------------------------------------------------------------------
#define LRG_MED_SMALL 1024      // lrg_median (stand-alone): up to 16 keys per lane, one wavefront per (slot, channel)
#define LRG_MED_LARGE 36864     // 144 KB: one workgroup per CU, only launched work for the few big regions
__device__ float lrg_median_wave(const LrgChanSrc cs, const int32_t *idx, int nc) {
    if (nc <= 256) return lrg_median_wave_r<4>(cs.base, idx, cs.stride, nc);       // the common case: the median Area-5 region has 57 points
    return lrg_median_wave_r<16>(cs.base, idx, cs.stride, nc);
}

__global__ __launch_bounds__(256) void lrg_median_wave_kernel(const LrgSlot *slots, const LrgRoom *rooms, LrgGrowParams prm,
                                                               float *center, int n_slots) {
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int s = pair >> 4, ch = pair & 15;
    if (s >= n_slots) return;
    const LrgSlot *S = &slots[s];
    const int F = prm.feature_size;
    const bool centred = (ch < 2 || ch >= 6) && ch < F;                         // :243-247
    const bool active = S->status == LRG_ACTIVE && S->room >= 0;
    if (!centred || !active) {
        if (lrg_lane() == 0) center[s * 16 + ch] = 0.f;
        return;
    }
    const int nc = S->nc;
    if (nc > LRG_MED_SMALL) return;                                             // the block-level launch owns this slot
    float med = lrg_median_wave(lrg_chan_src(&rooms[S->room], ch < 2 ? ch : ch - 4, ch, F), S->cur_idx, nc);
    if (lrg_lane() == 0) center[s * 16 + ch] = med;
}

// Fused per-slot preparation of a step for the counter stream: medians of small regions (large ones come from
// lrg_median_block_kernel, launched just before), subset sampling, gather + centre   (test_region_grow.py:237-254).
// 9 wavefronts: one per centred channel for the medians; then element-wise (coalesced) gather of the sampled rows.
#define LRG_PREP_THREADS 576
__global__ __launch_bounds__(LRG_PREP_THREADS) void lrg_prepare_kernel(const LrgSlot *slots, const LrgRoom *rooms,
                                                                       LrgGrowParams prm, float *center, int32_t *sample_in,
                                                                       int32_t *sample_nb, float *inlier, float *neighbor,
                                                                       int32_t *gt_remove, int32_t *gt_add, int32_t *rows_in,
                                                                       int32_t *rows_nb, int32_t *tile_total) {
    __shared__ float sh_c[16];
    __shared__ int sh_src[2][1024];
    const int s = blockIdx.x;
    TRACE2(s, 0);
    const LrgSlot *S = &slots[s];
    const bool active = S->status == LRG_ACTIVE && S->room >= 0;
    if (rows_in && threadIdx.x == 0) {
        const int ri = active ? min(S->nc, prm.n_inlier) : 0, rn = active ? min(S->ne, prm.n_neighbor) : 0;
        rows_in[s] = ri;
        rows_nb[s] = rn;
        if (tile_total && active) {
            // reserve this slot's entries in the two live-tile lists (order among slots does not matter)
            const int ti = (ri + LRG_ROW_TILE - 1) / LRG_ROW_TILE, tn = (rn + LRG_ROW_TILE - 1) / LRG_ROW_TILE;
            const int cap_in = (int)gridDim.x * ((prm.n_inlier + LRG_ROW_TILE - 1) / LRG_ROW_TILE);
            int32_t *list_in = tile_total + 2, *list_nb = tile_total + 2 + cap_in;
            const int bi = atomicAdd(&tile_total[0], ti), bn = atomicAdd(&tile_total[1], tn);
            for (int t = 0; t < ti; ++t) list_in[bi + t] = s * 64 + t;
            for (int t = 0; t < tn; ++t) list_nb[bn + t] = s * 64 + t;
        }
    }
    if (!active) return;
    const LrgRoom *R = &rooms[S->room];
    const int F = prm.feature_size, nc = S->nc, ne = S->ne;
    const int wave = threadIdx.x >> 6, lane = lrg_lane();
    const int kin = min(prm.n_inlier, 1024), knb = min(prm.n_neighbor, 1024);
    const float *points = R->points;
    const int32_t *obj = R->obj_id;
    // ---- subset sampling (:237-240, :249-252): positions -> source indices (independent of the centre) ----
    const uint32_t k0 = prm.rng_seed, k1 = (uint32_t)R->room_id;
    for (int u = threadIdx.x; u < kin + knb; u += blockDim.x) {
        const int side = u >= kin, j = side ? u - kin : u;
        const int pos = (int)lrg_sample_position((uint32_t)j, (uint32_t)(side ? ne : nc), (uint32_t)(side ? knb : kin),
                                                 side ? LRG_PURPOSE_NEIGHBOR : LRG_PURPOSE_INLIER, (uint32_t)S->seed,
                                                 (uint32_t)S->restart, (uint32_t)S->step, k0, k1);
        const int src = (side ? S->cand_idx : S->cur_idx)[pos];
        sh_src[side][j] = src;
        if (side) {
            sample_nb[(long)s * prm.n_neighbor + j] = pos;
            if (gt_add) gt_add[(long)s * prm.n_neighbor + j] = obj ? (obj[src] == S->target) : 0;          // :230,:254
        } else {
            sample_in[(long)s * prm.n_inlier + j] = pos;
            if (gt_remove) gt_remove[(long)s * prm.n_inlier + j] = obj ? (obj[src] != S->target) : 0;      // :231,:248
        }
    }
    // ---- centre (:241) ----
    if (threadIdx.x < 16) sh_c[threadIdx.x] = 0.f;
    __syncthreads();
    TRACE2(s, 1);
    // the medians of every slot come from lrg_median_block_kernel, launched just before: one (slot, channel) workgroup each,
    // a single wavefront for regions up to 1024 points -- nine of them in parallel beat nine waves sharing this workgroup's
    // staging of the rows (26 -> 12 us for this kernel)
    if (threadIdx.x < 16)
        sh_c[threadIdx.x] = ((threadIdx.x < 2 || threadIdx.x >= 6) && threadIdx.x < F) ? center[s * 16 + threadIdx.x] : 0.f;
    __syncthreads();
    TRACE2(s, 2);
#if LRG_TRACE
    if (threadIdx.x == 0 && g_lrg_trace2) g_lrg_trace2[(long)s * 16 + 8] = nc;
#endif
    // ---- gather + centre (:242-254), element-wise so that loads and stores of a row are contiguous across lanes ----
    for (int side = 0; side < 2; ++side) {
        const int k = side ? knb : kin;
        float *out = side ? neighbor + (long)s * prm.n_neighbor * F : inlier + (long)s * prm.n_inlier * F;
        const int nel = k * F;
        for (int e0 = threadIdx.x; e0 < nel; e0 += 8 * blockDim.x) {       // 8 independent row loads in flight per thread
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = min(e0 + u * (int)blockDim.x, nel - 1);
                const int j = e / F, f = e - j * F;
                v[u] = __fsub_rn(points[(long)sh_src[side][j] * F + f], sh_c[f]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * (int)blockDim.x;
                if (e < nel) out[e] = v[u];
            }
        }
    }
    TRACE2(s, 3);
}

#define LRG_MED_REGS (48 * 1024)   // largest region whose keys fit the registers of one 1024-thread workgroup
// Regions of (min_points, LRG_MED_REGS] points: keys in registers, 4 KB of LDS (radix select), so the many workgroups that find
// nothing to do (most slots hold small regions) come and go two per CU instead of queueing for a 147 KB LDS allocation each.
__global__ __launch_bounds__(1024) void lrg_median_block_kernel(const LrgSlot *slots, const LrgRoom *rooms, LrgGrowParams prm,
                                                                 float *center, int min_points, int32_t *tile_total) {
    if (tile_total && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { tile_total[0] = 0; tile_total[1] = 0; }   // lrg_prepare (next launch) fills the lists
    __shared__ __attribute__((aligned(16))) int sh[LRG_SAMPLED_LDS_INTS(1024)];      // (>= LRG_RADIX_LDS_INTS(1))
    const int s = blockIdx.x;
    const LrgSlot *S = &slots[s];
    const int F = prm.feature_size;
    const int ch = lrg_centred_channel(blockIdx.y, F);
    if (ch < 0 || S->status != LRG_ACTIVE || S->room < 0) return;
    const int nc = S->nc;
    if (nc <= min_points || nc > LRG_MED_REGS) return;                         // wave-level code / large-region kernel
    const LrgRoom *R = &rooms[S->room];
    if (nc <= 1024) {                                                          // one wavefront, keys in registers, no barriers
        if (threadIdx.x >= 64) return;
        const float m = lrg_median_wave(lrg_chan_src(R, blockIdx.y, ch, F), S->cur_idx, nc);
        if (threadIdx.x == 0) center[s * 16 + ch] = m;
        return;
    }
    const LrgChanSrc cs = lrg_chan_src(R, blockIdx.y, ch, F);
    const int chs[1] = {0};
    float med[1];
    if (nc <= 4096) lrg_median_block_radix<4, 1024, 1>(cs.base, chs, S->cur_idx, cs.stride, nc, sh, med);
    else if (nc <= 16 * 1024) lrg_median_block_radix<16, 1024, 1>(cs.base, chs, S->cur_idx, cs.stride, nc, sh, med);
    else {
#if LRG_MED48_BISECT == 2
        med[0] = lrg_median_block_sampled<48>(cs.base, S->cur_idx, cs.stride, nc, sh);
#elif LRG_MED48_BISECT
        if (threadIdx.x < 64) sh[threadIdx.x] = threadIdx.x == 0 ? -1 : 0;
        __syncthreads();
        med[0] = lrg_median_block_regs<48>(cs.base, S->cur_idx, cs.stride, nc, sh);
#else
        lrg_median_block_radix<48, 1024, 1>(cs.base, chs, S->cur_idx, cs.stride, nc, sh, med);
#endif
    }
    if (threadIdx.x == 0) center[s * 16 + ch] = med[0];
}

// Regions above LRG_MED_REGS points (only rooms that large can hold one: launched when max_points says so)
__global__ __launch_bounds__(1024) void lrg_median_large_kernel(const LrgSlot *slots, const LrgRoom *rooms, LrgGrowParams prm,
                                                                 float *center) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cache[];      // [LRG_MED_LARGE] keys, then 64 ints of scratch
    int *sh = reinterpret_cast<int *>(cache + LRG_MED_LARGE);
    const int s = blockIdx.x;
    const LrgSlot *S = &slots[s];
    const int F = prm.feature_size;
    const int ch = lrg_centred_channel(blockIdx.y, F);
    if (ch < 0 || S->status != LRG_ACTIVE || S->room < 0) return;
    const int nc = S->nc;
    if (nc <= LRG_MED_REGS) return;
    const LrgRoom *R = &rooms[S->room];
    const LrgChanSrc cs = lrg_chan_src(R, blockIdx.y, ch, F);
    if (threadIdx.x < 64) sh[threadIdx.x] = threadIdx.x == 0 ? -1 : 0;         // sh[0] = 0xFFFFFFFF (min identity)
    __syncthreads();
    const bool cached = nc <= LRG_MED_LARGE;
    if (cached) {
        // gather 8 values per thread per round: the index load and the dependent feature load of the 8 are independent,
        // so their two global latencies are paid once per round instead of once per element
        const int32_t *idx = S->cur_idx;
        const float *pts = cs.base;
        const int F = cs.stride;
        for (int j0 = threadIdx.x; j0 < nc; j0 += 8 * blockDim.x) {
            int id[8];
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { int jj = j0 + u * blockDim.x; id[u] = jj < nc ? idx[jj] : 0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pts[(long)id[u] * F];
#pragma unroll
            for (int u = 0; u < 8; ++u) { int jj = j0 + u * blockDim.x; if (jj < nc) cache[jj] = lrg_f2key(v[u]); }
        }
        __syncthreads();
    }
    const int k2 = nc >> 1;
    const int k1 = (nc & 1) ? k2 : k2 - 1;
    uint32_t ka, kb;
    lrg_select2(cache, cached, cs.base, S->cur_idx, cs.stride, 0, nc, k1, k2, sh, &ka, &kb);
    float lo = lrg_key2f(ka), hi = lrg_key2f(kb);
    float med = (nc & 1) ? hi : __fmul_rn(__fadd_rn(lo, hi), 0.5f);
    if (threadIdx.x == 0) center[s * 16 + ch] = med;
}

// ------------------------------------------------------------------------------------------------
// counter-stream subset sampling   (test_region_grow.py:237-240, :249-252)
// ------------------------------------------------------------------------------------------------
__global__ void lrg_sample_kernel(const LrgSlot *slots, const LrgRoom *rooms, LrgGrowParams prm, int32_t *sample_in,
                                  int32_t *sample_nb) {
    const int s = blockIdx.x;
    const LrgSlot *S = &slots[s];
    if (S->status != LRG_ACTIVE || S->room < 0) return;
    const uint32_t k0 = prm.rng_seed, k1 = (uint32_t)rooms[S->room].room_id;
    const int side = blockIdx.y;
    const int k = side == 0 ? prm.n_inlier : prm.n_neighbor;
    const int n = side == 0 ? S->nc : S->ne;
    int32_t *out = (side == 0 ? sample_in + (long)s * prm.n_inlier : sample_nb + (long)s * prm.n_neighbor);
    for (int j = threadIdx.x; j < k; j += blockDim.x)
        out[j] = (int32_t)lrg_sample_position((uint32_t)j, (uint32_t)n, (uint32_t)k,
                                               side == 0 ? LRG_PURPOSE_INLIER : LRG_PURPOSE_NEIGHBOR,
                                               (uint32_t)S->seed, (uint32_t)S->restart, (uint32_t)S->step, k0, k1);
}

// ------------------------------------------------------------------------------------------------
// gather + centre   (test_region_grow.py:242-254)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lrg_gather_center_kernel(const LrgSlot *slots, const LrgRoom *rooms,
                                                                 LrgGrowParams prm, const int32_t *sample_in,
                                                                 const int32_t *sample_nb, const float *center,
                                                                 float *inlier, float *neighbor, int32_t *gt_remove,
                                                                 int32_t *gt_add, int32_t *rows_in, int32_t *rows_nb) {
    const int s = blockIdx.x, side = blockIdx.y;
    const LrgSlot *S = &slots[s];
    const bool active = S->status == LRG_ACTIVE && S->room >= 0;
    if (rows_in && blockIdx.z == 0 && threadIdx.x == 0) {
        if (side == 0) rows_in[s] = active ? min(S->nc, prm.n_inlier) : 0;
        else rows_nb[s] = active ? min(S->ne, prm.n_neighbor) : 0;
    }
    if (!active) return;
    const LrgRoom *R = &rooms[S->room];
    const int F = prm.feature_size;
    const int k = side == 0 ? prm.n_inlier : prm.n_neighbor;
    const int32_t *samp = side == 0 ? sample_in + (long)s * prm.n_inlier : sample_nb + (long)s * prm.n_neighbor;
    const int32_t *list = side == 0 ? S->cur_idx : S->cand_idx;
    float *out = side == 0 ? inlier + (long)s * prm.n_inlier * F : neighbor + (long)s * prm.n_neighbor * F;
    const float *c = center + s * 16;
    const int chunk = (int)(blockDim.x * gridDim.z);
    for (int e = blockIdx.z * blockDim.x + threadIdx.x; e < k * F; e += chunk) {
        int j = e / F, f = e - j * F;
        int src = list[samp[j]];
        float v = R->points[(long)src * F + f];
        out[e] = __fsub_rn(v, c[f]);       // center[] is 0 on the channels the reference leaves alone (z, room xyz)
        if (f == 0) {
            if (side == 0 && gt_remove) gt_remove[(long)s * prm.n_inlier + j] = R->obj_id ? (R->obj_id[src] != S->target) : 0;   // :231,:248
            if (side == 1 && gt_add) gt_add[(long)s * prm.n_neighbor + j] = R->obj_id ? (R->obj_id[src] == S->target) : 0;      // :230,:254
        }
    }
}

// ------------------------------------------------------------------------------------------------
// confidence, Bernoulli masks, voxel-set update   (test_region_grow.py:262-288)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lrg_conf(const float *logit2) {
    // scipy.special.softmax(x)[1] in float32: exp(x - max) / sum
    float l0 = logit2[0], l1 = logit2[1];
    float m = fmaxf(l0, l1);
    float e0 = expf(l0 - m), e1 = expf(l1 - m);
    return __fdiv_rn(e1, __fadd_rn(e0, e1));
}

__global__ __launch_bounds__(512) void lrg_mask_update_kernel(LrgSlot *slots, const LrgRoom *rooms, LrgGrowParams prm,
                                                               const float *inlier, const float *neighbor,
                                                               const float *center, const float *add_logits,
                                                               const float *rmv_logits, const int32_t *gt_remove,
                                                               const int32_t *gt_add, const uint8_t *add_mask,
                                                               const uint8_t *rmv_mask, const int32_t *sample_in,
                                                               const int32_t *sample_nb, int64_t *stats) {
    __shared__ int sh_upd;
    __shared__ int sh_acc[2];
    const int s = blockIdx.x;
    LrgSlot *S = &slots[s];
    if (S->status != LRG_ACTIVE || S->room < 0) return;
    if (threadIdx.x < 2) sh_acc[threadIdx.x] = 0;
    const LrgRoom *R = &rooms[S->room];
    const int F = prm.feature_size;
    const float res = prm.resolution;
    const float c0 = center[s * 16 + 0], c1 = center[s * 16 + 1];
    const uint32_t k0 = prm.rng_seed, k1 = (uint32_t)R->room_id;
    uint8_t *cur = S->cur;
    if (threadIdx.x == 0) sh_upd = 0;
    __syncthreads();
    // ---- add pass (:266,:270-273,:283-285) ----
    const bool have_acc = add_logits && rmv_logits && gt_add && gt_remove;
    for (int j = threadIdx.x; j < prm.n_neighbor; j += blockDim.x) {
        bool take;
        const long o = (long)s * prm.n_neighbor + j;
        if (have_acc) {                                                        // add_acc (learn_region_grow_util.py:174-175)
            const long lo = (sample_nb && S->ne < prm.n_neighbor) ? (long)s * prm.n_neighbor + sample_nb[o] : o;
            if ((add_logits[2 * lo + 1] > add_logits[2 * lo] ? 1 : 0) == (gt_add[o] != 0 ? 1 : 0)) atomicAdd(&sh_acc[0], 1);
        }
        if (prm.policy == 2) take = gt_add[o] != 0;
        else if (add_mask) take = add_mask[o] != 0;
        else {
            // logits exist only for the distinct leading rows when the set was padded: read the source row's
            const long lo = (sample_nb && S->ne < prm.n_neighbor) ? (long)s * prm.n_neighbor + sample_nb[o] : o;
            float conf = lrg_conf(add_logits + 2 * lo);
            if (prm.policy == 1) take = conf > 0.5f;
            else take = lrg_uniform01(lrg_rng_word((uint32_t)j, LRG_PURPOSE_ADD, (uint32_t)S->seed, (uint32_t)S->restart,
                                                   (uint32_t)S->step, k0, k1)) < conf;
        }
        if (take) {
            const float *p = neighbor + o * F;
            int vx = lrg_voxel_of(__fadd_rn(p[0], c0), res);      // :271-272: un-centre x,y then rint(/res)
            int vy = lrg_voxel_of(__fadd_rn(p[1], c1), res);
            int vz = lrg_voxel_of(p[2], res);
            int idx = lrg_hash_lookup(R->hash_keys, R->hash_vals, R->hash_mask, lrg_pack_voxel(vx, vy, vz));
            if (idx >= 0 && !cur[idx]) { cur[idx] = 1; sh_upd = 1; }
        }
    }
    __syncthreads();
    // ---- remove pass (:267,:274-277,:286-287) ----
    for (int j = threadIdx.x; j < prm.n_inlier; j += blockDim.x) {
        bool take;
        const long o = (long)s * prm.n_inlier + j;
        if (have_acc) {                                                        // remove_acc (:179-180)
            const long lo = (sample_in && S->nc < prm.n_inlier) ? (long)s * prm.n_inlier + sample_in[o] : o;
            if ((rmv_logits[2 * lo + 1] > rmv_logits[2 * lo] ? 1 : 0) == (gt_remove[o] != 0 ? 1 : 0)) atomicAdd(&sh_acc[1], 1);
        }
        if (prm.policy == 2) take = gt_remove[o] != 0;
        else if (rmv_mask) take = rmv_mask[o] != 0;
        else {
            const long lo = (sample_in && S->nc < prm.n_inlier) ? (long)s * prm.n_inlier + sample_in[o] : o;
            float conf = lrg_conf(rmv_logits + 2 * lo);
            if (prm.policy == 1) take = conf > 0.5f;
            else take = lrg_uniform01(lrg_rng_word((uint32_t)j, LRG_PURPOSE_RMV, (uint32_t)S->seed, (uint32_t)S->restart,
                                                   (uint32_t)S->step, k0, k1)) < conf;
        }
        if (take) {
            const float *p = inlier + o * F;
            int vx = lrg_voxel_of(__fadd_rn(p[0], c0), res);
            int vy = lrg_voxel_of(__fadd_rn(p[1], c1), res);
            int vz = lrg_voxel_of(p[2], res);
            int idx = lrg_hash_lookup(R->hash_keys, R->hash_vals, R->hash_mask, lrg_pack_voxel(vx, vy, vz));
            if (idx >= 0) cur[idx] = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        S->updated = sh_upd;
        S->pad = 0;
        S->step += 1;
        S->steps_total += 1;                                                    // :288
        S->acc_add = have_acc ? sh_acc[0] : -1;
        S->acc_rmv = have_acc ? sh_acc[1] : -1;
        if (stats) atomicAdd(reinterpret_cast<unsigned long long *>(&stats[2]), 1ULL);
    }
}

// ------------------------------------------------------------------------------------------------
// 1-NN fill-in of unlabeled points   (test_region_grow.py:308-316)
// distance = numpy.sum((P - p)**2, axis=1) in float32 with NumPy's pairwise order; first minimum wins
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lrg_np_sqdist(const float *a, const float *b, int F) {
    float t[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) {
        if (l < F) {
            float d = __fsub_rn(a[l], b[l]);
            t[l] = __fmul_rn(d, d);
        } else t[l] = 0.f;
    }
    if (F < 8) {
        float s = t[0];
        for (int l = 1; l < F; ++l) s = __fadd_rn(s, t[l]);
        return s;
    }
    float s = __fadd_rn(__fadd_rn(__fadd_rn(t[0], t[1]), __fadd_rn(t[2], t[3])),
                        __fadd_rn(__fadd_rn(t[4], t[5]), __fadd_rn(t[6], t[7])));
    for (int l = 8; l < F; ++l) s = __fadd_rn(s, t[l]);
    return s;
}

__global__ __launch_bounds__(256) void lrg_nn1_fill_kernel(const float *points, int n, int F, const int32_t *label_in,
                                                            int32_t *label_out) {
    __shared__ float me[16];
    __shared__ float rd[4];
    __shared__ int rj[4];
    const int i = blockIdx.x;
    const int li = label_in[i];
    if (li != 0) { if (threadIdx.x == 0) label_out[i] = li; return; }
    if (threadIdx.x < 16) me[threadIdx.x] = threadIdx.x < F ? points[(long)i * F + threadIdx.x] : 0.f;
    __syncthreads();
    float best = INFINITY; int bj = INT_MAX;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        if (label_in[j] == 0) continue;
        float d = lrg_np_sqdist(points + (long)j * F, me, F);
        if (d < best || (d == best && j < bj) || bj == INT_MAX) { best = d; bj = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float od = __shfl_xor(best, o); int oj = __shfl_xor(bj, o);
        if (oj != INT_MAX && (bj == INT_MAX || od < best || (od == best && oj < bj))) { best = od; bj = oj; }
    }
    if (lrg_lane() == 0) { rd[threadIdx.x >> 6] = best; rj[threadIdx.x >> 6] = bj; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (rj[w] != INT_MAX && (bj == INT_MAX || rd[w] < best || (rd[w] == best && rj[w] < bj))) { best = rd[w]; bj = rj[w]; }
        label_out[i] = bj == INT_MAX ? 0 : label_in[bj];
    }
}

// The tiled formulation (lrg_nn1_fill_ws): the unlabeled points are compacted into a list; a workgroup takes 64 of them (one
// per lane, the query row in registers) and a chunk of 256 candidate rows staged in LDS -- every lane of a wavefront reads the
// same candidate row at the same time (LDS broadcast), the four wavefronts split the chunk -- and publishes its best
// (distance, index) with one 64-bit atomicMin per query: the distance is a non-negative float, so (bits << 32 | index) orders
// exactly like "smallest distance, then smallest index" (numpy.argmin's first minimum).  Candidate rows are read once per 64
// queries instead of once per query.
#define LRG_NN1_Q 64
#ifndef LRG_NN1_C
#define LRG_NN1_C 256
#endif
#define LRG_NN1_TARGET_WGS 4096      // workgroups the search grid aims for (256 CUs x 8)
// Up to LRG_FILL_BATCH rooms per launch (blockIdx.z = the room): the rooms that finish during one free-running launch are filled in
// together -- three launches for all of them instead of four per room (their gaps and tails were a third of a 13 k-point room's 35 us).
#define LRG_FILL_BATCH 64
struct LrgFillBatchArgs {
    const float *points[LRG_FILL_BATCH];
    const int32_t *label_in[LRG_FILL_BATCH];
    int32_t *label_out[LRG_FILL_BATCH];
    int32_t *list[LRG_FILL_BATCH];
    unsigned long long *best[LRG_FILL_BATCH];
    int32_t n[LRG_FILL_BATCH];
    int32_t *counts;         // [LRG_FILL_BATCH] unlabeled points of each room (zero before the prep kernel)
};

__global__ __launch_bounds__(256) void lrg_nn1_prep_kernel(LrgFillBatchArgs B) {
    // the list of a room's unlabeled points (in any order: every query is searched for on its own) -- ONE atomic per workgroup on the room's counter (one per
    // point, even aggregated per wavefront, was 70 us for sixteen 45 k-point rooms: the same address 700 times per room)
    __shared__ int wsum[4], base_sh;
    const int job = blockIdx.z, n = B.n[job];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool un = i < n && B.label_in[job][i] == 0;
    if (i < n) B.best[job][i] = ~0ull;
    const unsigned long long m = __ballot(un);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        base_sh = tot ? atomicAdd(B.counts + job, tot) : 0;
    }
    __syncthreads();
    if (un) {
        int off = base_sh + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) off += wsum[w];
        B.list[job][off] = i;
    }
}

__global__ __launch_bounds__(256) void lrg_nn1_search_kernel(LrgFillBatchArgs B, int F) {
    __shared__ float rows[LRG_NN1_C * 13 + 64];
    __shared__ int lab[LRG_NN1_C];
    __shared__ unsigned long long part[4][LRG_NN1_Q];
    const int job = blockIdx.z, n = B.n[job];
    const float *points = B.points[job];
    const int32_t *label_in = B.label_in[job], *list = B.list[job];
    unsigned long long *best = B.best[job];
    const int U = B.counts[job];
    if ((int)blockIdx.x * LRG_NN1_Q >= U) return;                  // (the host does not know U: a fixed, small number of query
                                                                   //  columns, each looping over its share of the list)
    const int c0 = blockIdx.y * LRG_NN1_C;
    if (c0 >= n) return;                                           // (the grid covers the largest room of the batch)
    const int nc = min(LRG_NN1_C, n - c0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < nc * F; e += blockDim.x) rows[e] = points[(long)c0 * F + e];       // contiguous block of rows
    for (int r = threadIdx.x; r < nc; r += blockDim.x) lab[r] = label_in[c0 + r];
    const int per = (nc + 3) / 4;
    for (int q0 = blockIdx.x * LRG_NN1_Q; q0 < U; q0 += gridDim.x * LRG_NN1_Q) {
        const int qi = list[min(q0 + lane, U - 1)];
        float me[16];
#pragma unroll
        for (int l = 0; l < 16; ++l) me[l] = l < F ? points[(long)qi * F + l] : 0.f;
        __syncthreads();                                               // rows staged / part[] of the previous round consumed
        unsigned long long bk = ~0ull;
        const int r1 = min(nc, (wave + 1) * per);
#pragma unroll 4
        for (int r = wave * per; r < r1; ++r) {                                          // branch-free: four candidates in flight
            const float d = lrg_np_sqdist(rows + r * F, me, F);
            const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(c0 + r);
            bk = min(bk, lab[r] != 0 ? k : ~0ull);
        }
        part[wave][lane] = bk;
        __syncthreads();
        if (wave == 0 && q0 + lane < U) {
            bk = min(min(part[0][lane], part[1][lane]), min(part[2][lane], part[3][lane]));
            if (bk != ~0ull) atomicMin(&best[qi], bk);
        }
    }
}

// The same search with the feature count known at compile time and TWO candidate rows per step of a lane: the rows of a chunk are
// staged as pairs interleaved feature by feature ([pair][feature][2]), so that one 8-byte LDS read feeds one packed subtraction,
// multiplication and addition (v_pk_*_f32: two candidates per instruction) -- each candidate's sum still formed in
// lrg_np_sqdist's order, operation by operation, so the distances are the same bits.  (The generic kernel above spends a third of its
// instructions on `l < F` selects and issues one 4-byte LDS read and three scalar-width VALU operations per candidate and feature:
// profiles/r03_fill_kernels.txt.)
typedef float lrg_f2 __attribute__((ext_vector_type(2)));
template <int FT>
__device__ __forceinline__ lrg_f2 lrg_np_sqdist2(const float *pair_rows, const lrg_f2 (&me2)[FT]) {
    lrg_f2 t[FT];
#pragma unroll
    for (int l = 0; l < FT; ++l) {
        const lrg_f2 c = *reinterpret_cast<const lrg_f2 *>(pair_rows + 2 * l);
        const lrg_f2 d = c - me2[l];
        t[l] = d * d;
    }
    if constexpr (FT < 8) {
        lrg_f2 sum = t[0];
#pragma unroll
        for (int l = 1; l < FT; ++l) sum = sum + t[l];
        return sum;
    } else {
        lrg_f2 sum = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
#pragma unroll
        for (int l = 8; l < FT; ++l) sum = sum + t[l];
        return sum;
    }
}

// QL query rows per lane (registers).  Measured with two (a candidate pair read from LDS then feeds four distances): the 64 largest rooms of the Area-5 set
// 3.23 ms either way, 64 KITTI-shaped scenes 9.96-10.04 ms against 6.87-6.88 ms with one (profiles/r04_fill_ql_ab.txt) -- the loop is not waiting for its LDS
// broadcasts; per candidate pair and query it issues 38 packed subtract / multiply / add (NumPy's order: no FMA) plus ~16 operations for the two 64-bit
// (distance, index) keys and their minima, i.e. ~0.41 of the quoted fp32 vector peak at best.  One query per lane.
#ifndef LRG_NN1_QL
#define LRG_NN1_QL 1
#endif
template <int FT>
__global__ __launch_bounds__(256) void lrg_nn1_search_pairs_kernel(LrgFillBatchArgs B) {
    constexpr int QL = LRG_NN1_QL, QB = LRG_NN1_Q * QL;                  // queries per workgroup and round
    __shared__ __attribute__((aligned(16))) float rows[LRG_NN1_C * FT];      // [pair][feature][2]
    __shared__ int lab[LRG_NN1_C];
    __shared__ unsigned long long part[4][QB];
    const int job = blockIdx.z, n = B.n[job];
    const float *points = B.points[job];
    const int32_t *label_in = B.label_in[job], *list = B.list[job];
    unsigned long long *best = B.best[job];
    const int U = B.counts[job];
    if ((int)blockIdx.x * QB >= U) return;
    const int c0 = blockIdx.y * LRG_NN1_C;
    if (c0 >= n) return;                                           // (the grid covers the largest room of the batch)
    const int nc = min(LRG_NN1_C, n - c0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < LRG_NN1_C * FT; e += blockDim.x) {
        const int r = e / FT, l = e - r * FT;
        rows[(r >> 1) * 2 * FT + 2 * l + (r & 1)] = r < nc ? points[(long)c0 * FT + e] : 0.f;      // (contiguous block of rows; a chunk's tail: zeros, never labeled)
    }
    // lab[r]: 0 for a labeled candidate, all ones for one without a label (or past the chunk's end) -- OR-ed into the distance's bit pattern below
    for (int r = threadIdx.x; r < LRG_NN1_C; r += blockDim.x) lab[r] = (r < nc && label_in[c0 + r] != 0) ? 0 : -1;
    constexpr int PER = LRG_NN1_C / 2 / 4;                             // pairs per wavefront
    for (int q0 = blockIdx.x * QB; q0 < U; q0 += gridDim.x * QB) {
        int qi[QL];
        lrg_f2 me2[QL][FT];
#pragma unroll
        for (int q = 0; q < QL; ++q) {
            qi[q] = list[min(q0 + q * LRG_NN1_Q + lane, U - 1)];
#pragma unroll
            for (int l = 0; l < FT; ++l) { const float v = points[(long)qi[q] * FT + l]; me2[q][l] = lrg_f2{v, v}; }
        }
        __syncthreads();                                               // rows staged / part[] of the previous round consumed
        // A wavefront walks its candidates in index order: "smaller distance, then smaller index" (numpy.argmin's first minimum) is a strict
        // compare of the distances' bit patterns as unsigned integers (distances are >= 0, so their order is that of their patterns; NaN after
        // +inf, exactly as in the 64-bit (distance, index) keys the rounds are combined with) -- an OR, a compare, a minimum and a select per
        // candidate instead of building such a key and taking its 64-bit minimum.
        unsigned bd[QL];
        int bi[QL];
#pragma unroll
        for (int q = 0; q < QL; ++q) { bd[q] = 0xFFFFFFFFu; bi[q] = -1; }
        for (int p = wave * PER; p < (wave + 1) * PER; ++p) {
            const int2 lb = *reinterpret_cast<const int2 *>(&lab[2 * p]);
#pragma unroll
            for (int q = 0; q < QL; ++q) {
                const lrg_f2 d = lrg_np_sqdist2<FT>(rows + p * 2 * FT, me2[q]);      // (the QL queries' reads of the pair: the same addresses, merged by the compiler)
                const unsigned d0 = __float_as_uint(d.x) | (unsigned)lb.x, d1 = __float_as_uint(d.y) | (unsigned)lb.y;      // (no label: 0xFFFFFFFF, never smaller)
                bi[q] = d0 < bd[q] ? 2 * p : bi[q];
                bd[q] = min(bd[q], d0);
                bi[q] = d1 < bd[q] ? 2 * p + 1 : bi[q];
                bd[q] = min(bd[q], d1);
            }
        }
#pragma unroll
        for (int q = 0; q < QL; ++q)
            part[wave][q * LRG_NN1_Q + lane] = bi[q] >= 0 ? (((unsigned long long)bd[q] << 32) | (unsigned)(c0 + bi[q])) : ~0ull;
        __syncthreads();
        if (wave < QL && q0 + wave * LRG_NN1_Q + lane < U) {
            const int j = wave * LRG_NN1_Q + lane;
            const unsigned long long m = min(min(part[0][j], part[1][j]), min(part[2][j], part[3][j]));
            if (m != ~0ull) atomicMin(&best[list[q0 + j]], m);
        }
    }
}

// (Measured and not kept, profiles/r04_fill_grid_ab.txt: the candidate rows through the scalar cache -- s_load into scalar registers, broadcast operands of the packed
//  operations, two queries per lane, no LDS at all -- +2 % on the Area-5 set, -28 % on 100 k-point scenes: eight workgroups' chunks are 104 KB against a 16 KB scalar
//  cache, and a scalar load's result can only be waited for with lgkmcnt(0).)
__global__ void lrg_nn1_write_kernel(LrgFillBatchArgs B) {
    const int job = blockIdx.z, n = B.n[job];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t *label_in = B.label_in[job];
    const int li = label_in[i];
    if (li != 0) { B.label_out[job][i] = li; return; }
    const unsigned long long k = B.best[job][i];
    B.label_out[job][i] = k == ~0ull ? 0 : label_in[(int)(k & 0xFFFFFFFFull)];
}

#include "lrg_front.inl"
#include "lrg_async.inl"
#include "lrg_beam.inl"

// ------------------------------------------------------------------------------------------------
extern "C" {

int lrg_voxelize(const float *points, int n, int F, float resolution, int32_t *voxels, void *stream) {
    if (!points || !voxels || n < 0 || F < 3 || !(resolution > 0.f)) return LRG_EINVAL - 1;
    if (n == 0) return 0;
    hipLaunchKernelGGL(lrg_voxelize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, points, n, F,
                       resolution, voxels);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_bind_group(LrgSlot *slots, LrgRoom *rooms, int first_slot, int group_size, int room, int reset_room, int clear_masks,
                   void *stream) {
    if (!slots || !rooms || first_slot < 0 || group_size < 1) return LRG_EINVAL - 1;
    hipLaunchKernelGGL(lrg_bind_group_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, slots, rooms, first_slot, group_size, room,
                       reset_room, clear_masks);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_voxel_pack(const int32_t *voxels, int n, int ox, int oy, int oz, uint32_t *pvox, int32_t *overflow_flag, void *stream) {
    if (!voxels || !pvox || !overflow_flag || n < 0) return LRG_EINVAL - 1;
    if (n == 0) return 0;
    hipLaunchKernelGGL(lrg_voxel_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, voxels, n, ox, oy, oz, pvox,
                       overflow_flag);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_voxel_grid_build(const int32_t *voxels, int n, int ox, int oy, int oz, int gx, int gy, int gz, int32_t *grid, void *stream) {
    if (!voxels || !grid || n < 0 || gx <= 0 || gy <= 0 || gz <= 0) return LRG_EINVAL - 1;
    LRG_HIP_CHECK(hipMemsetAsync(grid, 0xFF, (size_t)gx * gy * gz * sizeof(int32_t), (hipStream_t)stream));
    if (n == 0) return 0;
    hipLaunchKernelGGL(lrg_voxel_grid_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, voxels, n, ox, oy, oz, gx, gy, gz,
                       grid);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_voxel_hash_build(const int32_t *voxels, int n, uint64_t *keys, int32_t *vals, int hash_mask, int32_t *dup_flag,
                         void *stream) {
    if (!voxels || !keys || !vals || !dup_flag || n < 0 || hash_mask < 0 || ((hash_mask + 1) & hash_mask) != 0 ||
        hash_mask + 1 < n)
        return LRG_EINVAL - 1;
    LRG_HIP_CHECK(hipMemsetAsync(keys, 0xFF, (size_t)(hash_mask + 1) * sizeof(uint64_t), (hipStream_t)stream));
    if (n == 0) return 0;
    hipLaunchKernelGGL(lrg_hash_build_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, voxels, n, keys,
                       vals, hash_mask, dup_flag);
    LRG_LAUNCH_CHECK();
    return 0;
}

static int check_params(const LrgGrowParams *p) {
    if (!p || p->feature_size < 3 || p->feature_size > 16 || p->n_inlier <= 0 || p->n_neighbor <= 0 ||
        p->restarts < 1 || p->group_size < 1 || !(p->resolution > 0.f) || p->policy < 0 || p->policy > 2 || p->scoring < 0 ||
        p->scoring > 1)
        return LRG_EINVAL - 20;
    return 0;
}

int lrg_bbox_stop(LrgSlot *slots, const LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                  void *stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!slots || !rooms || n_slots <= 0 || max_points <= 0) return LRG_EINVAL - 1;
    const int nchunk = (max_points + LRG_SCAN_CHUNK - 1) / LRG_SCAN_CHUNK;
    hipLaunchKernelGGL(lrg_bbox_scan_kernel, dim3(n_slots, nchunk), dim3(LRG_SCAN_THREADS), 0, (hipStream_t)stream, slots,
                       rooms);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_advance(LrgSlot *slots, LrgRoom *rooms, int n_slots, const LrgGrowParams *params, int64_t *stats, void *stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!slots || !rooms || n_slots <= 0 || n_slots % params->group_size != 0) return LRG_EINVAL - 1;
    hipLaunchKernelGGL(lrg_advance_kernel<false>, dim3(n_slots / params->group_size), dim3(LRG_SCAN_THREADS), 0, (hipStream_t)stream,
                       slots, rooms, n_slots, *params, stats);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_box_query(LrgSlot *slots, const LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                  void *stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!slots || !rooms || n_slots <= 0 || max_points <= 0) return LRG_EINVAL - 1;
    const int nchunk = (max_points + LRG_SCAN_CHUNK - 1) / LRG_SCAN_CHUNK;
    hipLaunchKernelGGL(lrg_box_count_kernel, dim3(n_slots, nchunk), dim3(LRG_SCAN_THREADS), 0, (hipStream_t)stream, slots,
                       rooms);
    LRG_LAUNCH_CHECK();
    hipLaunchKernelGGL(lrg_box_compact_kernel, dim3(n_slots, nchunk), dim3(LRG_SCAN_THREADS), 0, (hipStream_t)stream,
                       slots, rooms, *params);
    LRG_LAUNCH_CHECK();
    return 0;
}

// block-level medians: the register kernel always, the LDS-cache kernel only when a room can hold a region that large
static int launch_block_medians(const LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params, float *center,
                                int min_points, int32_t *tile_total, int max_points, hipStream_t st) {
    const int ncentred = params->feature_size <= 2 ? params->feature_size : params->feature_size <= 6 ? 2 : params->feature_size - 4;
    hipLaunchKernelGGL(lrg_median_block_kernel, dim3(n_slots, ncentred), dim3(1024), 0, st, slots, rooms, *params, center, min_points,
                       tile_total);
    LRG_LAUNCH_CHECK();
    if (max_points > LRG_MED_REGS) {
        const size_t lds_large = LRG_MED_LARGE * 4 + 256;
        static bool attr_done[LRG_MAX_DEVICES] = {};
        const int dev = lrg_current_device();
        if (!attr_done[dev]) {
            LRG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(lrg_median_large_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_large));
            attr_done[dev] = true;
        }
        hipLaunchKernelGGL(lrg_median_large_kernel, dim3(n_slots, ncentred), dim3(1024), lds_large, st, slots, rooms, *params, center);
        LRG_LAUNCH_CHECK();
    }
    return 0;
}

int lrg_median(const LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params, float *center,
               void *stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!slots || !rooms || !center || n_slots <= 0) return LRG_EINVAL - 1;
    hipLaunchKernelGGL(lrg_median_wave_kernel, dim3((n_slots * 16 + 3) / 4), dim3(256), 0, (hipStream_t)stream, slots, rooms,
                       *params, center, n_slots);
    LRG_LAUNCH_CHECK();
    return launch_block_medians(slots, rooms, n_slots, params, center, LRG_MED_SMALL, nullptr, INT_MAX, (hipStream_t)stream);
}

static int prepare_impl(const LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params, float *center,
                        int32_t *sample_in, int32_t *sample_nb, float *inlier, float *neighbor, int32_t *gt_remove, int32_t *gt_add,
                        int32_t *rows_in, int32_t *rows_nb, int32_t *tile_total, int max_points, void *stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!slots || !rooms || !center || !sample_in || !sample_nb || !inlier || !neighbor || n_slots <= 0) return LRG_EINVAL - 1;
    if ((rows_in == nullptr) != (rows_nb == nullptr)) return LRG_EINVAL - 2;
    if (params->n_inlier > 1024 || params->n_neighbor > 1024) return LRG_EINVAL - 3;
    if ((rc = launch_block_medians(slots, rooms, n_slots, params, center, 0, tile_total, max_points, (hipStream_t)stream))) return rc;
    hipLaunchKernelGGL(lrg_prepare_kernel, dim3(n_slots), dim3(LRG_PREP_THREADS), 0, (hipStream_t)stream, slots, rooms, *params, center,
                       sample_in, sample_nb, inlier, neighbor, gt_remove, gt_add, rows_in, rows_nb, tile_total);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_prepare(const LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params, float *center,
                int32_t *sample_in, int32_t *sample_nb, float *inlier, float *neighbor, int32_t *gt_remove, int32_t *gt_add,
                int32_t *rows_in, int32_t *rows_nb, int32_t *tile_total, void *stream) {
    return prepare_impl(slots, rooms, n_slots, params, center, sample_in, sample_nb, inlier, neighbor, gt_remove, gt_add, rows_in,
                        rows_nb, tile_total, INT_MAX, stream);     // room sizes unknown here: both median kernels
}

int lrg_sample(const LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params, int32_t *sample_in,
               int32_t *sample_nb, void *stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!slots || !rooms || !sample_in || !sample_nb || n_slots <= 0) return LRG_EINVAL - 1;
    hipLaunchKernelGGL(lrg_sample_kernel, dim3(n_slots, 2), dim3(256), 0, (hipStream_t)stream, slots, rooms, *params,
                       sample_in, sample_nb);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_gather_center(const LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params,
                      const int32_t *sample_in, const int32_t *sample_nb, const float *center, float *inlier,
                      float *neighbor, int32_t *gt_remove, int32_t *gt_add, int32_t *rows_in, int32_t *rows_nb,
                      void *stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!slots || !rooms || !sample_in || !sample_nb || !center || !inlier || !neighbor || n_slots <= 0)
        return LRG_EINVAL - 1;
    if ((rows_in == nullptr) != (rows_nb == nullptr)) return LRG_EINVAL - 2;
    hipLaunchKernelGGL(lrg_gather_center_kernel, dim3(n_slots, 2, 4), dim3(256), 0, (hipStream_t)stream, slots, rooms,
                       *params, sample_in, sample_nb, center, inlier, neighbor, gt_remove, gt_add, rows_in, rows_nb);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_mask_update(LrgSlot *slots, const LrgRoom *rooms, int n_slots, const LrgGrowParams *params, const float *inlier,
                    const float *neighbor, const float *center, const float *add_logits, const float *rmv_logits,
                    const int32_t *gt_remove, const int32_t *gt_add, const uint8_t *add_mask, const uint8_t *rmv_mask,
                    const int32_t *sample_in, const int32_t *sample_nb, int64_t *stats, void *stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!slots || !rooms || !inlier || !neighbor || !center || n_slots <= 0) return LRG_EINVAL - 1;
    if (params->policy == 2 && (!gt_remove || !gt_add)) return LRG_EINVAL - 2;
    if (params->policy != 2 && ((add_mask == nullptr) != (rmv_mask == nullptr))) return LRG_EINVAL - 3;
    if (params->policy != 2 && !add_mask && (!add_logits || !rmv_logits)) return LRG_EINVAL - 4;
    hipLaunchKernelGGL(lrg_mask_update_kernel, dim3(n_slots), dim3(512), 0, (hipStream_t)stream, slots, rooms, *params,
                       inlier, neighbor, center, add_logits, rmv_logits, gt_remove, gt_add, add_mask, rmv_mask, sample_in,
                       sample_nb, stats);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_grow_step(LrgSlot *slots, LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                  const LrgWeights *weights, const LrgStepBuffers *b, int advance_rounds, unsigned forward_flags,
                  void *stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!slots || !rooms || !weights || !b || n_slots <= 0 || advance_rounds < 1) return LRG_EINVAL - 1;
    if (weights->feature_size != params->feature_size) return LRG_EINVAL - 2;
    if (params->scoring == 1 && params->restarts > 1) return LRG_EINVAL - 7;    // the 'ml' score is accumulated by lrg_grow_step_packed only
    const bool fuse_scan = params->group_size == 1 && max_points <= 65536;     // (one workgroup scanning a 100 k-point scene is slower
                                                                               //  than the chunked launch)
    if (fuse_scan) {
        // greedy growing: the scan of the updated mask rides in the advance kernel (one workgroup per slot either way)
        if (n_slots % params->group_size != 0) return LRG_EINVAL - 1;
        hipLaunchKernelGGL(lrg_advance_kernel<true>, dim3(n_slots), dim3(LRG_SCAN_THREADS), 0, (hipStream_t)stream, slots, rooms,
                           n_slots, *params, b->stats);
        LRG_LAUNCH_CHECK();
        if ((rc = lrg_box_query(slots, rooms, n_slots, max_points, params, stream))) return rc;
    } else if ((rc = lrg_bbox_stop(slots, rooms, n_slots, max_points, params, stream))) {
        return rc;
    }
    for (int r = fuse_scan ? 1 : 0; r < advance_rounds; ++r) {
        if ((rc = lrg_advance(slots, rooms, n_slots, params, b->stats, stream))) return rc;
        if ((rc = lrg_box_query(slots, rooms, n_slots, max_points, params, stream))) return rc;
    }
    const bool rows = b->rows_in && b->rows_nb && (forward_flags & LRG_FWD_FUSED);
    int32_t *tile_lists = nullptr;
    if (rows && params->n_inlier <= 64 * LRG_ROW_TILE && params->n_neighbor <= 64 * LRG_ROW_TILE) {
        size_t off = 0, cnt = 0;
        if ((rc = lrg_forward_workspace_view(weights, n_slots, params->n_inlier, params->n_neighbor, 6, 0, &off, &cnt))) return rc;
        tile_lists = reinterpret_cast<int32_t *>(static_cast<float *>(b->workspace) + off);
        forward_flags |= LRG_FWD_TILE_LISTS;
    }
    if ((rc = prepare_impl(slots, rooms, n_slots, params, b->center, b->sample_in, b->sample_nb, b->inlier, b->neighbor,
                           b->gt_remove, b->gt_add, rows ? b->rows_in : nullptr, rows ? b->rows_nb : nullptr, tile_lists,
                           max_points, stream)))
        return rc;
    if ((rc = lrg_forward_rows(weights, b->inlier, b->neighbor, n_slots, params->n_inlier, params->n_neighbor,
                               rows ? b->rows_in : nullptr, rows ? b->rows_nb : nullptr, b->add_logits, b->rmv_logits,
                               b->workspace, b->workspace_bytes, forward_flags, stream))) return rc;
    return lrg_mask_update(slots, rooms, n_slots, params, b->inlier, b->neighbor, b->center, b->add_logits,
                           b->rmv_logits, b->gt_remove, b->gt_add, nullptr, nullptr, rows ? b->sample_in : nullptr,
                           rows ? b->sample_nb : nullptr, b->stats, stream);
}

static bool lrg_uses_greedy_front(const LrgGrowParams *params, const LrgPackedBuffers *b) {
    return params->group_size == 1 && params->restarts == 1 && b->rooms_have_pvox && b->slot_big && params->n_inlier <= 512 &&
           params->n_neighbor <= 512;
}

const float *lrg_packed_rows_center(const LrgGrowParams *params, const LrgPackedBuffers *b) {
    return (params && b && lrg_uses_greedy_front(params, b)) ? b->center : nullptr;
}

static int front_step_impl(LrgSlot *slots, LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                           const LrgWeights *weights, const LrgPackedBuffers *b, void *stream, bool launch_medians) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!slots || !rooms || !weights || !b || n_slots <= 0 || max_points <= 0) return LRG_EINVAL - 1;
    if (weights->feature_size != params->feature_size) return LRG_EINVAL - 2;
    if (n_slots % params->group_size != 0) return LRG_EINVAL - 1;
    if (max_points > LRG_FRONT_MAXCHUNK * LRG_SCAN_CHUNK) return LRG_EINVAL - 3;       // larger rooms: lrg_grow_step
    if (params->n_inlier > LRG_FRONT_MAXSAMPLE || params->n_neighbor > LRG_FRONT_MAXSAMPLE) return LRG_EINVAL - 3;
    if (!b->center || !b->sample_in || !b->sample_nb || !b->x_in || !b->x_nb || !b->row_slot_in || !b->row_slot_nb || !b->upd_in ||
        !b->upd_nb || !b->rmv_logits || !b->add_logits || !b->slot_rows || !b->counters || !b->workspace)
        return LRG_EINVAL - 4;
    if (b->row_cap % LRG_ROW_TILE != 0 || (long)b->row_cap < (long)n_slots * LRG_PAD_ROWS(max(params->n_inlier, params->n_neighbor)))
        return LRG_EINVAL - 5;
    size_t poff = 0, pcnt = 0;
    if ((rc = lrg_forward_packed_pooled_view(weights, n_slots, b->row_cap, &poff, &pcnt))) return rc;
    if (b->workspace_bytes < lrg_forward_packed_workspace_bytes(weights, n_slots, b->row_cap)) return LRG_EINVAL - 6;
    LrgFrontArgs a = {};
    a.center = b->center; a.sample_in = b->sample_in; a.sample_nb = b->sample_nb;
    a.x_in = b->x_in; a.x_nb = b->x_nb; a.row_slot_in = b->row_slot_in; a.row_slot_nb = b->row_slot_nb;
    a.upd_in = reinterpret_cast<float4 *>(b->upd_in); a.upd_nb = reinterpret_cast<float4 *>(b->upd_nb); a.rmv_logits = b->rmv_logits; a.add_logits = b->add_logits;
    a.slot_rows = b->slot_rows; a.counters = b->counters;
    a.pooled = static_cast<float *>(b->workspace) + poff; a.pooled_stride = (int)(pcnt / (size_t)n_slots);
    a.stats = b->stats;
    a.phase_ticks = b->phase_ticks;
    a.own_medians = 0; a.row_stride = 0; a.phase_dbg = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (lrg_uses_greedy_front(params, b)) {
        const int ncentred = params->feature_size <= 2 ? params->feature_size : params->feature_size <= 6 ? 2 : params->feature_size - 4;
        hipLaunchKernelGGL(lrg_front_greedy_kernel, dim3(n_slots), dim3(LRG_FRONT_THREADS), 0, st, slots, rooms, n_slots, *params, a, b->slot_big);
        LRG_LAUNCH_CHECK();
        if (launch_medians) {
            hipLaunchKernelGGL(lrg_front_big_kernel, dim3(n_slots, ncentred), dim3(1024), 0, st, slots, rooms, *params, a, b->slot_big);
            LRG_LAUNCH_CHECK();
        }
    } else if (params->group_size == 1) {
        hipLaunchKernelGGL(lrg_front_kernel<7>, dim3(n_slots), dim3(LRG_FRONT_THREADS), 0, st, slots, rooms, n_slots, *params, a);
        LRG_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(lrg_front_kernel<1>, dim3(n_slots), dim3(LRG_FRONT_THREADS), 0, st, slots, rooms, n_slots, *params, a);
        LRG_LAUNCH_CHECK();
        hipLaunchKernelGGL(lrg_advance_kernel<false>, dim3(n_slots / params->group_size), dim3(LRG_SCAN_THREADS), 0, st, slots, rooms,
                           n_slots, *params, b->stats);
        LRG_LAUNCH_CHECK();
        hipLaunchKernelGGL(lrg_front_kernel<4>, dim3(n_slots), dim3(LRG_FRONT_THREADS), 0, st, slots, rooms, n_slots, *params, a);
        LRG_LAUNCH_CHECK();
    }
    return 0;
}

int lrg_front_step(LrgSlot *slots, LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                   const LrgWeights *weights, const LrgPackedBuffers *b, void *stream) {
    return front_step_impl(slots, rooms, n_slots, max_points, params, weights, b, stream, true);
}

int lrg_grow_step_packed(LrgSlot *slots, LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                         const LrgWeights *weights, const LrgPackedBuffers *b, void *stream) {
    if (!params || !b) return LRG_EINVAL - 1;
    int rc = front_step_impl(slots, rooms, n_slots, max_points, params, weights, b, stream, true);
    if (rc) return rc;
    return lrg_forward_packed(weights, b->x_in, b->x_nb, lrg_packed_rows_center(params, b), b->row_slot_in, b->row_slot_nb, b->counters,
                              b->counters + 2, n_slots, b->row_cap, b->add_logits, b->rmv_logits, b->workspace, b->workspace_bytes,
                              LRG_FWD_POOL_ZEROED, stream);
}


static size_t async_ring_entries(int n_slots) {
    size_t ring = 1 << 14;                                   // entries: far more than the tasks that can be outstanding (~80 per slot)
    while (ring < (size_t)n_slots * 512) ring <<= 1;
    return ring;
}
static size_t async_unit_ring_entries(int n_slots) {         // the pooled-product units' ring: a slot has one entry outstanding at most -- and with batched pooled
    size_t ring = 256;                                       // products (LRG_GEMV_BATCH) a closed batch takes LRG_GEMV_BATCH positions whatever it holds: live batches
    while (ring < (size_t)n_slots * 2 * LRG_GEMV_BATCH) ring <<= 1;      // span up to LRG_GEMV_BATCH x n_slots positions (twice that: nobody wraps onto an unread entry)
    return ring;
}
static size_t async_wave_ring_entries(int n_slots) {         // one of the eight (side, quarter) rings of a wave-branch launch: a slot has at most 16 tiles per side outstanding
    size_t ring = 2048;
    while (ring < (size_t)n_slots * 32) ring <<= 1;
    return ring;
}
size_t lrg_grow_async_queue_bytes(int n_slots) {
    if (n_slots <= 0) return 0;
    // (two task rings: branch tiles | pooled blocks and head tiles; then the units' ring; then the fill-in ring; then the wave rings' control words and the eight wave rings)
    return (LRG_AQ_RING + 2 * async_ring_entries(n_slots) + async_unit_ring_entries(n_slots) + LRG_ASYNC_FILL_RING + 256 + 8 * async_wave_ring_entries(n_slots)) * sizeof(int32_t);
}

// the network lrg_wave_tile.inl is written for: 9 .. 16 features, branch layers 64, 64, 64, 128, 512 (the LrgNet of the paper, lite 0)
static bool lrg_wave_branch_fits(const LrgWeights *w) {
    return w->feature_size > 8 && w->feature_size <= LRG_WB_K0 && w->n_conv == 5 && w->conv_ch[0] == LRG_WB_C0 && w->conv_ch[1] == LRG_WB_C1 && w->conv_ch[2] == LRG_WB_C2 &&
           w->conv_ch[3] == LRG_WB_C3 && w->conv_ch[4] == LRG_WB_C4;
}

// the side stream and the two events of a wave-branch launch (per device; created once)
// (a ring of event pairs: a launch's events are not recorded again while an earlier launch's waits on them may still be queued -- callers run a few launches ahead)
#define LRG_SIDE_EVENTS 64
struct LrgSideStream { hipStream_t stream; hipEvent_t start[LRG_SIDE_EVENTS], done[LRG_SIDE_EVENTS]; unsigned next; bool ok; };
static LrgSideStream *lrg_side_stream() {
    static LrgSideStream side[LRG_MAX_DEVICES] = {};
    LrgSideStream *s = &side[lrg_current_device()];
    if (!s->ok) {
        // HIP streams are mapped onto a few hardware queues (four per priority level), round robin by creation, and kernels of two streams that share a queue never run
        // side by side: with a side stream of the callers' own priority every fourth new caller stream landed on its queue -- the worker kernel waited out its 4 s for a
        // front kernel queued BEHIND it (3 of 12 growers, tools/r06_spec_soak.py).  The side stream is created at the HIGHEST priority: a queue of another pool than any
        // stream of default priority.  (A caller that launches from a highest-priority stream of its own can still collide: found out by the start rendezvous, reason 6 / 2.)
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if (hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, prio_greatest) != hipSuccess) return nullptr;
        for (int i = 0; i < LRG_SIDE_EVENTS; ++i)
            if (hipEventCreateWithFlags(&s->start[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->done[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        s->ok = true;
    }
    return s;
}

size_t lrg_grow_async_tail_bytes(int n_slots, int tail_rows) {
    if (n_slots <= 0 || tail_rows <= 0 || (tail_rows & 31)) return 0;
    // (the two row cursors on a 64-byte line each, two words per shared tile and side, the slots' tail bases)
    return (size_t)(32 + 4 * (size_t)(tail_rows / 32) + 2 * (size_t)n_slots) * sizeof(int32_t);
}

size_t lrg_grow_async_pool_rows_bytes(const LrgWeights *weights, int n_slots) {
    if (!weights || n_slots <= 0 || weights->n_conv < 1) return 0;
    return (size_t)n_slots * 2 * 16 * (size_t)weights->conv_ch[weights->n_conv - 1] * sizeof(float);      // [slot][side][tile][columns of the pooled layer]
}

int lrg_grow_async(LrgSlot *slots, LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params, const LrgWeights *weights,
                   const LrgPackedBuffers *b, const LrgAsyncBuffers *ab, int max_steps, int budget_us, void *stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!slots || !rooms || !weights || !b || !ab || n_slots <= 0 || max_points <= 0 || max_steps < 1 || budget_us < 0) return LRG_EINVAL - 1;
    if (weights->feature_size != params->feature_size) return LRG_EINVAL - 2;
    if (!lrg_uses_greedy_front(params, b) || max_points > LRG_FRONT_MAXCHUNK * LRG_SCAN_CHUNK) return LRG_EINVAL - 3;
    if (!b->center || !b->sample_in || !b->sample_nb || !b->x_in || !b->x_nb || !b->row_slot_in || !b->row_slot_nb || !b->upd_in ||
        !b->upd_nb || !b->rmv_logits || !b->add_logits || !b->slot_rows || !b->counters || !b->workspace || !ab->queue || !ab->sync)
        return LRG_EINVAL - 4;
    const int row_stride = (max(params->n_inlier, params->n_neighbor) + 31) / 32 * 32;
    if (b->row_cap % LRG_ROW_TILE != 0 || (long)b->row_cap < (long)n_slots * row_stride) return LRG_EINVAL - 5;
    const size_t qbytes = lrg_grow_async_queue_bytes(n_slots);
    if (ab->queue_bytes < qbytes || ((uintptr_t)ab->queue & 255) || n_slots >= (1 << 20)) return LRG_EINVAL - 6;
    size_t poff = 0, pcnt = 0;
    if ((rc = lrg_forward_packed_pooled_view(weights, n_slots, b->row_cap, &poff, &pcnt))) return rc;

    LrgAsyncKArgs K;
    K.slots = slots; K.rooms = rooms; K.prm = *params;
    LrgAsyncArgs &A = K.A;
    LrgFusedArgs branches, heads;
    if ((rc = lrg_packed_problems(weights, b->x_in, b->x_nb, b->center, b->row_slot_in, b->row_slot_nb, b->counters, n_slots, b->row_cap,
                                  b->add_logits, b->rmv_logits, b->workspace, b->workspace_bytes, &branches, &A.gemv, &heads)))
        return rc;
    A.prob[0] = branches.p[0]; A.prob[1] = branches.p[1]; A.prob[2] = heads.p[0]; A.prob[3] = heads.p[1];
    for (int i = 0; i < 4; ++i) {
        const LrgFusedProb &P = A.prob[i];
        // the shapes the team tiles are instantiated for: lite 0 / 2 (lite 1 stores conv[1] from the accumulators: lrg_grow_step_packed)
        const int Kp = (P.Kin + 7) & ~7;
        if (32 * (Kp + 4) > (i < 2 ? 32 * 132 : 32 * 68)) return LRG_EINVAL - 7;
        for (int l = 0; l < P.nlayers; ++l) {
            const LrgFusedLayer &L = P.L[l];
            const bool inplace = (L.flags & LRG_FL_INPLACE) != 0, to_buf1 = ((l & 1) != 0) == !inplace;
            if (L.gout && (!(L.flags & LRG_FL_KEEP) || inplace)) return LRG_EINVAL - 7;
            if ((L.flags & LRG_FL_KEEP) && 32 * (L.N + 4) > (to_buf1 ? (i < 2 ? 32 * 132 : 32 * 68) : (i < 2 ? 32 * 68 : 32 * 260))) return LRG_EINVAL - 7;
        }
    }
    if (A.gemv.P + 8 * LRG_GEMV_TASK_COLS > LRG_ASYNC_TILE_FLOATS || (A.gemv.P & 127) || (A.gemv.ldw & 3) || (A.gemv.C & 3) ||
        (((uintptr_t)A.gemv.w[0] | (uintptr_t)A.gemv.w[1]) & 15))
        return LRG_EINVAL - 7;                              // (pooled product: sixteen 16-byte rows in flight per lane, K ranges of P / 8)
    LrgFrontArgs &a = A.front;
    a.center = b->center; a.sample_in = b->sample_in; a.sample_nb = b->sample_nb;
    a.x_in = b->x_in; a.x_nb = b->x_nb; a.row_slot_in = b->row_slot_in; a.row_slot_nb = b->row_slot_nb;
    a.upd_in = reinterpret_cast<float4 *>(b->upd_in); a.upd_nb = reinterpret_cast<float4 *>(b->upd_nb); a.rmv_logits = b->rmv_logits; a.add_logits = b->add_logits;
    a.slot_rows = b->slot_rows; a.counters = b->counters;
    a.pooled = static_cast<float *>(b->workspace) + poff; a.pooled_stride = (int)(pcnt / (size_t)n_slots);
    a.stats = b->stats;
    a.phase_ticks = nullptr;
    a.own_medians = 1; a.row_stride = row_stride;
    a.rows16 = 0;
    if (ab->rows16 && params->feature_size <= 16 && params->feature_size > 8 && (((uintptr_t)b->x_in | (uintptr_t)b->x_nb | (uintptr_t)b->center) & 15) == 0) {
        // the caller's row arrays hold row_cap x 16 floats: gathered rows at a 64-byte stride, written and read in 16-byte pieces
        a.rows16 = 1;
        A.prob[0].ldx = 16; A.prob[1].ldx = 16;
    }
    a.phase_dbg = ab->debug_ticks ? reinterpret_cast<unsigned long long *>(ab->debug_ticks) + 20 : nullptr;

    hipDeviceProp_t prop;
    LRG_HIP_CHECK(hipGetDeviceProperties(&prop, lrg_current_device()));
    int wgs = prop.multiProcessorCount;
    if (ab->compute_units > 0 && ab->compute_units < wgs) wgs = ab->compute_units;
    // Front workgroups: one per two slots (a front step takes ~23 us of a ~85 us step, and a CU a front workgroup holds is a CU without
    // tile teams: 68 slots, two teams: 34 / 40 / 46 / 68 front workgroups 806 / 810 / 807 / 792 k instance-steps/s, profiles/r03_units_sweep.log);
    // one per slot while the slots are few and CUs plenty (eight 100 k-point scenes: 8 / 4 front workgroups 97 k / 86 k, profiles/r03_kitti2_*.json)
    // ... and never more than an eighth of the CUs (+ 2: 34 of 256) by default: a front step takes ~24 us whatever the number of slots, so
    // ~1 M steps/s keep ~25 front workgroups busy, and every CU beyond that is a CU without tile teams (192 slots: 24 / 28 / 34 / 48 front
    // workgroups 0.92 / 1.06 / 1.17 / 1.11 M instance-steps/s; 272 slots: 34 / 40 / 48 / 64 / 96: 1.11 / 1.07 / 1.02 / 0.94 / 0.76 M)
    // (with shared tail tiles -- a sixth fewer branch tiles -- the tile teams need fewer CUs and the slots more: 272 slots 34 / 40 / 46 / 52 front workgroups
    //  828 (without them) / 872 / 871 / 858 rooms/s, 320 slots 40 / 44 / 48: 876 / 886 / 881, profiles/r05_tail_fronts*.txt)
    const bool tails_on = ab->tail_ctl && ab->tail_rows > 0 && ab->rows16 && !ab->pool_rows;
    int n_front = ab->front_workgroups > 0 ? ab->front_workgroups : n_slots <= 24 ? n_slots : min((n_slots + 1) / 2, tails_on ? wgs * 11 / 64 : wgs / 8 + 2);
    n_front = min(n_front, n_slots);
    n_front = max(n_front, (n_slots + LRG_ASYNC_MAX_SERVED - 1) / LRG_ASYNC_MAX_SERVED);
    n_front = min(n_front, wgs / 2);                         // (at least half of the CUs for the tile teams)
    if (n_front < (n_slots + LRG_ASYNC_MAX_SERVED - 1) / LRG_ASYNC_MAX_SERVED) return LRG_EINVAL - 8;      // more slots than the front workgroups can serve
    // Speculation (LrgAsyncBuffers.speculate = K > 1): groups of K slots on one room each, one front workgroup per group (a group's shared state -- the
    // room's visited flags, labels, seed cursor, the other slots' boxes -- stays on one CU and needs no hand-over)
    a.spec_k = 0;
    a.spec_stats = nullptr;
    if (ab->speculate > 1) {
        if (ab->speculate > LRG_ASYNC_MAX_SERVED || n_slots % ab->speculate != 0 || n_slots / ab->speculate > wgs / 2) return LRG_EINVAL - 8;
        a.spec_k = ab->speculate;
        n_front = n_slots / ab->speculate;
        if (ab->work) a.spec_stats = reinterpret_cast<unsigned long long *>(ab->work) + 4;
    }
    // Pooled-product units (lrg_async.inl): sixteen CUs for the LrgNet of the paper (2 heads x 256 columns, 1024 pooled features).
    // Off (-1), or where the slices do not fit / would leave the tile teams fewer than half of the CUs: the teams' 128-column blocks.  Off
    // above 176 slots too: sixteen units take ~1.1 M pooled products a second, and the head tiles that wait for them hold their teams
    // (136 / 160 / 192 / 272 slots with | without units: 1.13 | 1.03, 1.15 | 1.11, 1.14 | 1.17, 1.06 | 1.11 M instance-steps/s, profiles/r03_slots_sweep.log;
    //  end of round 4, profiles/r04_teams_units_sweep.txt: 136 / 160 / 176 slots 1.15 | 1.09, 1.16 | 1.19, 1.17 | 1.23 M -- off above 148).
    {
        const LrgGemvArgs &g = A.gemv;
        int units = (g.C % LRG_GEMV_UNIT_COLS == 0) ? 2 * g.C / LRG_GEMV_UNIT_COLS : 0;
        // (register-tile launches keep the units up to their last slot count: 160 slots with units and every fourth worker CU on two branch teams 1.26 M against 1.20 M
        //  instance-steps/s for the one-kernel launch without units, profiles/r06_reg_tiles_slots.txt)
        static const int wave_env0 = getenv("LRG_ASYNC_WAVES") ? atoi(getenv("LRG_ASYNC_WAVES")) : 0;
        const int want0 = wave_env0 ? wave_env0 : ab->branch_waves;
        const bool reg_candidate = (want0 == 1 || (want0 == 0 && n_slots >= LRG_REG_TILE_AUTO_MIN && n_slots <= LRG_REG_TILE_AUTO_MAX)) && ab->rows16 && !tails_on && !ab->pool_rows &&
                                   lrg_wave_branch_fits(weights) && !(ab->compute_units > 0 && ab->compute_units < prop.multiProcessorCount);
        if (ab->gemv_units < 0 || (ab->gemv_units == 0 && n_slots > (reg_candidate ? LRG_REG_TILE_AUTO_MAX : 148)) || (size_t)max((int)LRG_GEMV_UNIT_FLOATS(g.P), (int)LRG_GEMV_UNIT2_FLOATS(g.P)) * sizeof(float) + 16 > 160 * 1024 || n_slots > LRG_GEMV_UNIT_MAX_SLOTS || (((uintptr_t)g.pooled) & 15) || (g.P & 127) ||
            n_front + units > wgs / 2 + wgs / 4 || n_slots >= (1 << 20))
            units = 0;
        A.gemv_units = units;
        // (half-teams in the units, lrg_async_gemv_unit2: eight tasks in flight per unit, each a little longer.  68 / 100 / 136 slots with register tiles: 992 -> 975,
        //  1 147 -> 1 189, 1 156 -> 1 206 k instance-steps/s (profiles/r06_unit_pairs.txt): on from 84 slots, where the units' capacity is what a slot queues for;
        //  LRG_ASYNC_UNIT_PAIRS=0 / 1 forces)
        A.unit_pairs = getenv("LRG_ASYNC_UNIT_PAIRS") ? (atoi(getenv("LRG_ASYNC_UNIT_PAIRS")) ? 1 : 0) : (units && n_slots >= 84 ? 1 : 0);
        // without the units: the pooled products in batches (lrg_async.inl, LRG_GEMV_BATCH) where slots become ready faster than a batch's patience --
        // LRG_ASYNC_GEMV_BATCH=0 / =1: off / on whatever the slot count; LRG_ASYNC_GEMV_BATCH_US: the patience
        const int batch_env = getenv("LRG_ASYNC_GEMV_BATCH") ? atoi(getenv("LRG_ASYNC_GEMV_BATCH")) : -1;
        const int batch_us10 = getenv("LRG_ASYNC_GEMV_BATCH_US") ? (int)(10.0 * atof(getenv("LRG_ASYNC_GEMV_BATCH_US"))) : 15;
        const bool fits = (size_t)(LRG_GEMV_BATCH * (g.P + 4)) <= (size_t)LRG_ASYNC_TILE_FLOATS && g.P == 1024 && (g.C & 31) == 0 && n_slots < LRG_GEMV_NOBODY;      // (half ranges of eight k-groups: the paper's 2 x 512 pooled features)
        // (2 176 room jobs, rooms/s without | with: 200 slots 771 | 731, 272: 873 | 800, 320: 888 | 850, 400: 880 | 896 -- a block takes 68 us for ~7.7 slots instead
        //  of 23 us for one, 35 instead of 93 team-us per evaluation, but the pooled stage of a slot's step grows from 71 to 108 us and below ~400 slots the launch is
        //  bound by that latency, not by its teams: profiles/r05_gemv_batch_v1.txt, r05_bench_debug_batch.log)
        // ... and the matrix-core form of the block (this build): 200 slots 773 | 723, 272: 871 | 791, 320: 885 | 837, 400: 884 | 877; with shared head tiles 400: 884-911 | 864,
        // 544: 906 | 905 (profiles/r05_gemv_batch_v3_mfma_pipelined.txt, r05_tail_heads2.txt).  The block for one slot costs the tile teams' CU nothing but L2 latency
        // (its few FMAs run beside the other teams' MFMAs); the batch's block costs matrix-pipe time -- 32-row tiles for ~8 slots -- which is what the launch is short of.
        // Off unless asked for (LRG_ASYNC_GEMV_BATCH=1).
        A.gemv_batch = (units == 0 && fits && batch_env > 0) ? LRG_GEMV_BATCH : 0;
        A.gemv_batch_ticks = (long long)batch_us10 * 10;
    }
    // With the units, a branch tile leaves its column maxima of the pooled layer as one row of 16-byte stores (pool_rows) and the units take
    // the maximum over a slot's tiles while loading: no atomicMax per column (7 k atomics = write transactions per evaluation, each to
    // be acknowledged before the tile may report in), no zeroing of the pooled feature by the front workgroup.
    // In-launch fill-in (lrg_async.inl): the caller's arenas for the lists of unlabeled points and the best (distance, index) words, laid out like
    // the label arena; 13 features (the compiled-in search); one team each of up to sixteen worker workgroups serves the fill-in ring.
    A.fill_list = nullptr; A.fill_best = nullptr; A.fill_sync = nullptr; A.fill_label_base = nullptr; A.fill_out_base = nullptr; A.fill_wgs = 0; A.fill_extra = 0; A.fill_hybrid = 0;
    a.fill_in_launch = 0;
    if (ab->fill_list && ab->fill_best && ab->fill_sync && ab->fill_label_base && ab->fill_out_base && params->feature_size == 13 && ab->fill_rooms > 0 &&
        ab->fill_rooms <= (1 << 18) && max_points <= 1024 * LRG_NN1_C) {
        if ((uintptr_t)ab->fill_best & 7) return LRG_EINVAL - 6;
        A.fill_list = ab->fill_list; A.fill_best = reinterpret_cast<unsigned long long *>(ab->fill_best); A.fill_sync = ab->fill_sync;
        A.fill_label_base = ab->fill_label_base; A.fill_out_base = ab->fill_out_base;
        a.fill_in_launch = 1;
    }
    // Shared tail tiles: the caller's row arrays continue behind the slots' own rows
    A.tail = nullptr; A.tail_tiles = 0; A.tail_ticks = 0; A.tail_heads = 0;
    a.tail_cur = nullptr; a.tail_base = nullptr; a.tail_rows = 0; a.tail_row0 = 0;
    if (ab->tail_ctl && ab->tail_rows > 0 && a.rows16 && !ab->pool_rows) {
        // (+ 32 rows: a slot's own head tile on its tail rows stages a full tile from the tail's first row, tail_heads == 0)
        if ((ab->tail_rows & 31) || ((uintptr_t)ab->tail_ctl & 63) || (long)b->row_cap < (long)n_slots * row_stride + ab->tail_rows + 32 || n_slots >= (1 << 20) ||
            ab->tail_rows / 32 >= (1 << 20))
            return LRG_EINVAL - 9;
        A.tail = ab->tail_ctl; A.tail_tiles = ab->tail_rows / 32;
        A.tail_ticks = ab->tail_close_us < 0 ? 0 : ab->tail_close_us > 0 ? (long long)ab->tail_close_us * 100 : 200;
        a.tail_cur = ab->tail_ctl; a.tail_rows = ab->tail_rows; a.tail_row0 = n_slots * row_stride;
        a.tail_base = ab->tail_ctl + 32 + 4 * (size_t)A.tail_tiles;
        // (the heads of the tails on the shared tiles too -- without the units, whose head tiles start before the pooled product is complete and wait inside;
        //  LRG_ASYNC_TAIL_HEADS=0: a head tile of the slot's own per tail)
        A.tail_heads = (A.gemv_units == 0 && !(getenv("LRG_ASYNC_TAIL_HEADS") && atoi(getenv("LRG_ASYNC_TAIL_HEADS")) == 0)) ? 1 : 0;
        LRG_HIP_CHECK(hipMemsetAsync(ab->tail_ctl, 0, (32 + 4 * (size_t)A.tail_tiles) * sizeof(int32_t), (hipStream_t)stream));
    }
    A.pool_rows = nullptr; A.pool_rows_stride = 0;
    if (A.gemv_units && ab->pool_rows && row_stride <= 512 && n_slots <= 4096) {
        const size_t need = (size_t)n_slots * 2 * 16 * (A.gemv.P / 2) * sizeof(float);
        if (ab->pool_rows_bytes < need || ((uintptr_t)ab->pool_rows & 15)) return LRG_EINVAL - 6;
        A.pool_rows = ab->pool_rows; A.pool_rows_stride = 2 * 16 * (A.gemv.P / 2);
        for (int side = 0; side < 2; ++side) {
            A.prob[side].pool_rows = ab->pool_rows + (size_t)side * 16 * (A.gemv.P / 2);
            A.prob[side].pool_rows_stride = A.pool_rows_stride;
        }
        a.pooled = nullptr;          // (nobody accumulates into it)
    }
    // Two-kernel launches (round 6, lrg_wave_tile.inl / lrg_grow_async_worker_kernel): the tile CUs as a second kernel of 512 threads and up to 256 VGPRs, resident
    // beside the front workgroups' and units' kernel.  LrgAsyncBuffers.branch_waves: 1 = REGISTER TILES (a branch tile by a team of four wavefronts, layers 0 - 2 per
    // wavefront in registers, one barrier: the default from LRG_REG_TILE_AUTO_MIN to LRG_REG_TILE_AUTO_MAX slots), 4 / 8 = one-wavefront PREFIX / POOL tasks on CUs that
    // keep the kernels of their stage in LDS, -1 = one kernel.  Needs the rows at a 64-byte stride, the paper's network, no shared tail tiles / per-tile pool rows, and the
    // whole chip (the two grids are sized per shader engine: a CU-masked launch keeps the one-kernel form).
    A.wave_wgs = 0; A.wave_a_wgs = 0; A.wave_waves = 0; A.wave_split = 4; A.wave_fill = 0; A.wmask = (int)async_wave_ring_entries(n_slots) - 1; A.h3[0] = A.h3[1] = nullptr; A.reg_tiles = 0;
    // (every fourth register-tile CU with two branch teams from 120 slots: a slot's branch tiles queue for their teams there -- 68 slots -2.5 %, 100: +0.3 %, 136: +3.6 %, 160: +4.6 %)
    A.rt_bb_every = getenv("LRG_ASYNC_RT_BB_EVERY") ? atoi(getenv("LRG_ASYNC_RT_BB_EVERY")) : (n_slots >= 120 ? 4 : 0);
    int worker_wgs = 0;                                      // workgroups of the worker kernel (wave-branch mode)
    {
        static const int wave_env = getenv("LRG_ASYNC_WAVES") ? atoi(getenv("LRG_ASYNC_WAVES")) : 0;
        int want = wave_env ? wave_env : ab->branch_waves;
        if (want == 0 && n_slots >= LRG_REG_TILE_AUTO_MIN && n_slots <= LRG_REG_TILE_AUTO_MAX) want = 1;      // (register tiles where they win: include/lrg_hip.h)
        const bool can = a.rows16 && !A.tail && !A.pool_rows && !A.gemv_batch && lrg_wave_branch_fits(weights) &&      // (batched pooled products: tasks of the one-kernel launch's teams)
                         A.prob[0].nlayers == 5 && A.prob[0].L[1].gout && A.prob[0].pool &&
                         (ab->compute_units <= 0 || ab->compute_units >= prop.multiProcessorCount) && (wgs % 32) == 0 && wgs >= 64 && n_slots < (1 << 20);
        if (can && want > 0) {
            // Both kernels' workgroups go round the 8 XCDs in turn, and inside an XCD round its 4 shader engines (8 CUs each) -- a workgroup whose engine has no CU
            // free WAITS for one instead of going elsewhere, and where a kernel's round starts depends on what was dispatched before.  So the grids are sized per
            // shader engine for ANY alignment of the two rounds: the front kernel's F = n_front + units workgroups put at most f = ceil(ceil(F / 8) / 4) on one
            // engine, the worker kernel may then have 8 - f per engine: W = 32 x (8 - f).  (Sized per XCD only -- 21 + 232 workgroups -- 7 of 10 launches gave up at the
            // start rendezvous with every workgroup arrived in the end: the last worker workgroups had waited for CUs that front workgroups held, tools/r06_wave_rendezvous.py;
            // tools/two_kernel_rendezvous.hip, profiles/r03_side_stream: "a kernel of another stream is only placed when EVERY shader engine has a CU to spare".)
            // Front workgroups are added while the engines they already claim have room and there are slots for them.
            const int engines = 32, per_engine = wgs / engines;
            static const int fronts_env = getenv("LRG_ASYNC_WAVE_FRONTS") ? atoi(getenv("LRG_ASYNC_WAVE_FRONTS")) : 0;
            if (ab->front_workgroups <= 0 && !a.spec_k && fronts_env > 0) n_front = min(fronts_env, n_slots);
            int f = ((n_front + A.gemv_units + 7) / 8 + 3) / 4;
            if (ab->front_workgroups <= 0 && !a.spec_k && fronts_env <= 0) n_front = max(n_front, min(n_slots, engines * f - A.gemv_units));
            const int F = n_front + A.gemv_units;
            f = ((F + 7) / 8 + 3) / 4;
            worker_wgs = engines * (per_engine - f);
            // (test hook: that many worker workgroups MORE than the shader engines hold -- a launch that can never be resident as a whole: its front workgroups
            //  give up at the start rendezvous with reason 6, the workgroups that start after that find the abort word and leave;
            //  tests/test_gpu_free_run.py::test_wave_branch_launch_that_cannot_be_resident_gives_up_cleanly)
            if (getenv("LRG_ASYNC_WAVE_EXTRA_WGS")) worker_wgs += max(0, atoi(getenv("LRG_ASYNC_WAVE_EXTRA_WGS")));
            static const int wwgs_env = getenv("LRG_ASYNC_WAVE_WGS") ? atoi(getenv("LRG_ASYNC_WAVE_WGS")) : 0;
            static const int awgs_env = getenv("LRG_ASYNC_WAVE_A_WGS") ? atoi(getenv("LRG_ASYNC_WAVE_A_WGS")) : 0;
            static const int split_env = getenv("LRG_ASYNC_WAVE_SPLIT") ? atoi(getenv("LRG_ASYNC_WAVE_SPLIT")) : 0;
            // (per evaluation ~7 tiles: PREFIX tasks ~7 x 10 us of one wavefront, POOL tasks ~28 x 9 us, four wavefronts to a CU; head tiles ~7 x 13-17 us of a team, two
            //  to a CU -- and the pooled blocks where there are no units: 62 % | 55 % of the worker CUs run branch tasks, a fifth of those the PREFIX tasks)
            int wave_wgs = wwgs_env > 0 ? wwgs_env : worker_wgs * (A.gemv_units ? 62 : 55) / 100;
            wave_wgs = max(8, min(wave_wgs / 4 * 4, worker_wgs - 8));
            int a_wgs = awgs_env > 0 ? awgs_env : (wave_wgs + 2) / 5;
            a_wgs = max(1, min(a_wgs, wave_wgs - 4));
            a_wgs += (wave_wgs - a_wgs) % 4;                 // (POOL CUs: a multiple of four -- the (side, half) kinds)
            size_t c3[2];
            if (want == 1) {
                // REGISTER TILES (LrgAsyncBuffers.branch_waves = 1): every worker workgroup alike -- team 0 the branch tiles (four wavefronts per tile, activations in
                // registers where that is free: lrg_team_branch_tile_reg), team 1 the pooled blocks and head tiles
                if (worker_wgs >= 24) A.reg_tiles = (getenv("LRG_ASYNC_RT_TEAM_HEADS") && atoi(getenv("LRG_ASYNC_RT_TEAM_HEADS"))) ? 2 : 1;      // (2: register branch tiles, team head tiles)
            } else if (worker_wgs >= 24 && lrg_packed_conv3_view(weights, n_slots, b->row_cap, c3) == 0) {
                A.wave_wgs = wave_wgs; A.wave_a_wgs = a_wgs;
                A.wave_waves = want > 0 ? min(want, 8) : 4;
                A.wave_split = split_env == 8 ? 8 : 4;
                A.h3[0] = static_cast<float *>(b->workspace) + c3[0]; A.h3[1] = static_cast<float *>(b->workspace) + c3[1];
            }
        }
    }
    // Teams per worker CU: two -- the first runs branch tiles, the second head tiles (a tile beside another takes 1.2 x as long, but
    // the head tiles wait inside for the pooled-product units, and at 68 slots the teams are what a step queues for: 1 / 2 / 3 teams
    // 751 / 806 / 771 k instance-steps/s with 34 front workgroups, profiles/r03_units_sweep.log); one while the slots are few (nothing
    // queues, a tile alone is faster: eight scenes 108 k against 97 k); three where hundreds of slots are in flight
    const int teams = ab->teams > 0 ? min(ab->teams, 4) : n_slots <= 24 ? 1 : n_slots <= 96 ? (A.gemv_units ? 2 : 1) : n_slots <= (A.gemv_units ? 128 : 200) ? 3 : 4;      // (112 slots: 2 / 3 teams 1.02 / 1.04 M; 96: 1.00 / 0.96 M)
    A.queue = ab->queue; A.sync = ab->sync; A.big = b->slot_big; A.room_queue = ab->room_queue; A.work = reinterpret_cast<unsigned long long *>(ab->work); A.dbg = reinterpret_cast<unsigned long long *>(ab->debug_ticks);
    A.qmask = (int)async_ring_entries(n_slots) - 1;
    A.gmask = (int)async_unit_ring_entries(n_slots) - 1;
    const bool two_kernels = A.wave_wgs || A.reg_tiles;
    A.n_slots = n_slots; A.n_front = n_front; A.teams = two_kernels ? 2 : teams;
    A.worker_base = n_front + A.gemv_units; A.total_wgs = two_kernels ? n_front + A.gemv_units + worker_wgs : wgs;
    if (A.fill_list && A.reg_tiles) {
        // both teams of a register-tile CU run tiles: the second team of the first fill_wgs of them serves the fill-in ring instead of ring 1 (LRG_ASYNC_RT_FILL_WGS / LrgAsyncBuffers.fill_wgs).
        // Default 0: the host fills finished rooms in between launches -- 68 rooms in flight, 0 / 8 / 16 / 32 such workgroups: 931 / 892 / 918 / 920 k instance-steps/s
        // (profiles/r06_reg_tiles_sweep.txt): a head team less per CU costs more than the fill-ins between two launches
        static const int rt_fill_env = getenv("LRG_ASYNC_RT_FILL_WGS") ? atoi(getenv("LRG_ASYNC_RT_FILL_WGS")) : -1;
        const int want_fill = ab->fill_wgs > 0 ? ab->fill_wgs : rt_fill_env >= 0 ? rt_fill_env : 0;
        A.fill_wgs = min(want_fill, worker_wgs / 4);
        if (A.fill_wgs < 1) { A.fill_list = nullptr; a.fill_in_launch = 0; }
    } else if (A.fill_list && A.wave_wgs) {
        // the fill-in teams: wavefronts 4 .. 7 of wave-branch CUs (VALU work beside the MFMA-bound branch wavefronts of the same SIMDs), unless those run branch tasks too
        if (A.wave_waves <= 4) {
            A.wave_fill = 1;
            A.fill_wgs = ab->fill_wgs > 0 ? min(ab->fill_wgs, A.wave_wgs) : min(64, A.wave_wgs);
        } else {
            A.fill_list = nullptr; a.fill_in_launch = 0;      // (the host fills in between launches)
        }
    } else if (A.fill_list) {
        const int workers = wgs - n_front - A.gemv_units;
        // (16 / 32 / 64 / 103 such workgroups at 68 rooms in flight: 852 / 858 / 857 / 857 k instance-steps/s, 6: 804 k -- a big room's ~170 tasks queue for
        //  them; without the in-launch fill-in 852 k: profiles/r04_fill_in_launch_ab.txt)
        A.fill_wgs = ab->fill_wgs > 0 ? min(ab->fill_wgs, workers / 2) : min(64, workers / 3);      // (end of round 4, 2 176 rooms: 32 / 64 / 96 such workgroups 581 / 587 / 587 rooms/s at 68 slots, 839 / 856 / 844 at 272)
        if (A.fill_wgs < 1) { A.fill_list = nullptr; a.fill_in_launch = 0; }      // (too few workgroups: the host fills in)
        A.fill_extra = (A.fill_list && teams <= 3) ? 1 : 0;      // (a fourth team of 256 threads beside three tile teams; its LDS region is 16 KB)
        // four tile teams: the fill-in team is the CU's fourth tile team and serves ring 1 while no fill-in task waits (LRG_ASYNC_FILL_HYBRID=0: the fill-in ring only)
        static const int hybrid_env = getenv("LRG_ASYNC_FILL_HYBRID") ? atoi(getenv("LRG_ASYNC_FILL_HYBRID")) : 1;
        A.fill_hybrid = (A.fill_list && teams == 4 && hybrid_env) ? 1 : 0;
    }
    {
        static const int r0_env = getenv("LRG_ASYNC_RING0_HALVES") ? atoi(getenv("LRG_ASYNC_RING0_HALVES")) : 0;
        A.ring0_halves = r0_env > 0 ? r0_env : teams >= 3 ? 3 : 2;
        A.head_ring = (teams > 1 && (r0_env >= 0 || teams == 4)) ? 1 : 0;      // (LRG_ASYNC_RING0_HALVES=-1: one ring)
        // four teams: 2 x (branch tile: 28 KB) + 2 x (head tile: 44.5 KB) = 145 KB of the CU's 160; the first two run branch tiles only
        static const int small_env = getenv("LRG_ASYNC_SMALL_TEAMS") ? atoi(getenv("LRG_ASYNC_SMALL_TEAMS")) : 0;      // 2, 3, or 23 = 2 / 3 on even / odd workgroups
        A.small_teams = teams == 4 ? (small_env == 3 ? 3 : 2) : 0;
        A.small_alt = (teams == 4 && small_env == 23) ? 1 : 0;
    }
    A.poll_sleep = ab->poll_sleep > 0 ? ab->poll_sleep : 1;
    // few slots, most teams idle: a branch tile as two tasks that share its pooled layer (tile 22.8 -> 18.4 us; eight 100 k-point scenes
    // 75.9 k -> 78.1 k instance-steps/s, four tasks 75.0 k; 68 rooms: 559 k -> 505 k, the teams are busy there: profiles/r03_parts_perf.log)
    A.branch_parts = ab->branch_parts > 0 ? (ab->branch_parts >= 4 ? 4 : ab->branch_parts >= 2 ? 2 : 1) : (n_slots <= 46 ? 2 : 1);      // (end of round 4, profiles/r04_teams_units_sweep.txt: 16 / 24 / 39 / 44 / 52 / 68 slots, 2 against 1 part: +8 / +6 / +2.3 / +1.5 / -2 / -17 %)
    if (A.wave_wgs) { A.branch_parts = A.wave_split; A.head_ring = 1; A.small_teams = 0; A.small_alt = 0; A.fill_extra = 0; }
    if (A.reg_tiles) {
        // (a register branch tile as two tasks where CUs idle: LrgAsyncBuffers.branch_parts >= 2 / LRG_ASYNC_RT_PARTS, by default up to 24 slots)
        static const int rt_parts_env = getenv("LRG_ASYNC_RT_PARTS") ? atoi(getenv("LRG_ASYNC_RT_PARTS")) : 0;
        A.branch_parts = rt_parts_env > 0 ? (rt_parts_env >= 2 ? 2 : 1) : ab->branch_parts > 0 ? (ab->branch_parts >= 2 ? 2 : 1) : (n_slots <= 24 ? 2 : 1);
        A.head_ring = 1; A.small_teams = 0; A.small_alt = 0; A.fill_extra = 0;
    }      // (a branch tile = its four quarters; ring 1 for everything else)
    A.max_steps = max_steps;
    A.budget_ticks = budget_us > 0 ? (long long)budget_us * 100 : (1LL << 60);      // wall_clock64: 100 MHz
    A.abort_ticks = (budget_us > 0 ? (long long)budget_us * 100 : 0) + 400000000LL;  // ... + 4 s without an end: something is broken
    A.start_ticks = ab->start_wait_us > 0 ? (long long)ab->start_wait_us * 100 : (budget_us > 0 ? (long long)budget_us * 100 : 0) + LRG_ASYNC_START_TICKS;
    static_assert(sizeof(LrgAsyncKArgs) <= 4096, "kernel arguments");
    hipStream_t st = (hipStream_t)stream;
    LRG_HIP_CHECK(hipMemsetAsync(ab->queue, 0, qbytes, st));
    LRG_HIP_CHECK(hipMemsetAsync(ab->sync, 0, (size_t)n_slots * LRG_ASYNC_SYNC_WORDS * sizeof(int32_t), st));
    const size_t front_lds = ((sizeof(LrgFrontShared) + 15) & ~(size_t)15) + sizeof(LrgAsyncFrontCtl);
    // (small_alt: the odd workgroups have one small team more and one big team less -- the even ones' layout is the larger)
    const size_t team_lds = ((size_t)A.small_teams * LRG_ASYNC_SMALL_TEAM_FLOATS + (size_t)(teams - A.small_teams) * LRG_ASYNC_TEAM_FLOATS +
                             (size_t)A.fill_extra * LRG_ASYNC_FILL_TEAM_FLOATS) * sizeof(float);
    static_assert((3 * LRG_ASYNC_TEAM_FLOATS + LRG_ASYNC_FILL_TEAM_FLOATS) * sizeof(float) <= 160 * 1024, "three tile teams and a fill team per CU");
    static_assert((2 * LRG_ASYNC_SMALL_TEAM_FLOATS + 2 * LRG_ASYNC_TEAM_FLOATS) * sizeof(float) <= 160 * 1024, "four tile teams per CU");
    const size_t unit_lds = A.gemv_units ? (size_t)max((int)LRG_GEMV_UNIT_FLOATS(A.gemv.P), (int)LRG_GEMV_UNIT2_FLOATS(A.gemv.P)) * sizeof(float) + 16 : 0;
    const size_t lds = (max(max(front_lds, two_kernels ? (size_t)0 : team_lds), unit_lds) + 15) & ~(size_t)15;
    // (wave-branch mode, the worker kernel: a wave-branch CU's kernels + its fill-in team | two tile teams)
    const size_t worker_lds = (max(max((size_t)(LRG_WB_FLOATS + LRG_ASYNC_FILL_TEAM_FLOATS), (size_t)2 * LRG_ASYNC_TEAM_FLOATS),
                                   (size_t)(LRG_RT_WEIGHT_FLOATS + LRG_RT_TEAM0_FLOATS + max((int)LRG_ASYNC_TEAM_FLOATS, (int)LRG_RT_TEAM1_FLOATS))) * sizeof(float) + 15) & ~(size_t)15;
    static_assert((LRG_WB_FLOATS + LRG_ASYNC_FILL_TEAM_FLOATS) * sizeof(float) <= 160 * 1024, "a wave-branch CU: the kernels of its (side, quarter) and a fill-in team");
    static bool attr_done[LRG_MAX_DEVICES] = {};
    const int dev = lrg_current_device();
    if (!attr_done[dev]) {
        LRG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(lrg_grow_async_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        LRG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(lrg_grow_async_worker_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[dev] = true;
    }
    // Residency: front workgroups, units and tile teams wait for each other, so the launch is only correct when ALL its workgroups run at
    // once -- one per CU.  Checked here instead of found out by a spin bound seconds later: the kernel as compiled must fit a CU with this
    // much LDS, and the stream must be allowed at least `wgs` CUs (a CU-masked stream, hipExtStreamCreateWithCUMask, is allowed fewer).
    // What cannot be seen from here (another process or stream holding CUs) is caught by the launch's own start rendezvous within
    // LRG_ASYNC_START_TICKS (lrg_async.inl: abort reason 6), not by the hand-overs' multi-second bounds.
    {
        int per_cu = 0;
        LRG_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(lrg_grow_async_kernel), LRG_FRONT_THREADS, lds));
        if (per_cu < 1) return LRG_ERESIDENCY;
        if (two_kernels) {
            LRG_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(lrg_grow_async_worker_kernel), LRG_WORKER_THREADS, worker_lds));
            if (per_cu < 1) return LRG_ERESIDENCY;
        }
        uint32_t cumask[32] = {};
        const uint32_t words = (uint32_t)min(32, (prop.multiProcessorCount + 31) / 32);
        if (hipExtStreamGetCUMask(st, words, cumask) == hipSuccess) {
            int visible = 0;
            for (uint32_t i = 0; i < words; ++i) visible += __builtin_popcount(cumask[i]);
            if (visible > 0 && visible < (two_kernels ? prop.multiProcessorCount : wgs)) return LRG_ERESIDENCY;      // (no bit set: no mask reported)
        } else {
            (void)hipGetLastError();
        }
    }
    if (two_kernels) {
        // Two kernels, resident together: the worker kernel on the side stream between two events of the caller's stream (it starts after everything the caller
        // enqueued before this call -- the memsets above included -- and the caller's stream goes on only when it has left), the front kernel on the caller's stream.
        LrgSideStream *side = lrg_side_stream();
        if (!side) return LRG_ERESIDENCY;
        LrgAsyncKArgs KW = K;
        KW.A.worker_base = A.wave_wgs;                       // (the tile teams' workgroups are numbered from the first one behind the wave-branch CUs)
        const unsigned ev = side->next++ % LRG_SIDE_EVENTS;
        LRG_HIP_CHECK(hipEventRecord(side->start[ev], st));
        LRG_HIP_CHECK(hipStreamWaitEvent(side->stream, side->start[ev], 0));
        hipLaunchKernelGGL(lrg_grow_async_worker_kernel, dim3(worker_wgs), dim3(LRG_WORKER_THREADS), worker_lds, side->stream, KW);
        LRG_LAUNCH_CHECK();
        LRG_HIP_CHECK(hipEventRecord(side->done[ev], side->stream));
        hipLaunchKernelGGL(lrg_grow_async_kernel, dim3(n_front + A.gemv_units), dim3(LRG_FRONT_THREADS), lds, st, K);
        LRG_LAUNCH_CHECK();
        LRG_HIP_CHECK(hipStreamWaitEvent(st, side->done[ev], 0));
        return 0;
    }
    hipLaunchKernelGGL(lrg_grow_async_kernel, dim3(wgs), dim3(LRG_FRONT_THREADS), lds, st, K);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_beam_advance(LrgBeamGroup *groups, LrgSlot *slots, LrgRoom *rooms, int n_groups, int beam_width, int search_width,
                     const LrgGrowParams *params, int64_t *stats, void *stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!groups || !slots || !rooms || n_groups <= 0 || beam_width < 1 || beam_width > LRG_BEAM_MAXQ || search_width < 1 ||
        beam_width * search_width > 64)
        return LRG_EINVAL - 1;
    hipLaunchKernelGGL(lrg_beam_advance_kernel, dim3(n_groups), dim3(1024), 0, (hipStream_t)stream, groups, slots, rooms, beam_width,
                       search_width, *params, stats);
    LRG_LAUNCH_CHECK();
    return 0;
}

int lrg_beam_level(LrgBeamGroup *groups, LrgSlot *slots, LrgRoom *rooms, int n_groups, int beam_width, int search_width, int max_points,
                   const LrgGrowParams *params, const LrgWeights *weights, const LrgStepBuffers *b, unsigned forward_flags, void *stream) {
    int rc = lrg_beam_advance(groups, slots, rooms, n_groups, beam_width, search_width, params, b ? b->stats : nullptr, stream);
    if (rc) return rc;
    if (!weights || !b || max_points <= 0) return LRG_EINVAL - 1;
    const int n_slots = n_groups * beam_width * search_width;
    if ((rc = lrg_box_query(slots, rooms, n_slots, max_points, params, stream))) return rc;
    const bool rows = b->rows_in && b->rows_nb && (forward_flags & LRG_FWD_FUSED);
    int32_t *tile_lists = nullptr;
    if (rows && params->n_inlier <= 64 * LRG_ROW_TILE && params->n_neighbor <= 64 * LRG_ROW_TILE) {
        size_t off = 0, cnt = 0;
        if ((rc = lrg_forward_workspace_view(weights, n_slots, params->n_inlier, params->n_neighbor, 6, 0, &off, &cnt))) return rc;
        tile_lists = reinterpret_cast<int32_t *>(static_cast<float *>(b->workspace) + off);
        forward_flags |= LRG_FWD_TILE_LISTS;
    }
    if ((rc = prepare_impl(slots, rooms, n_slots, params, b->center, b->sample_in, b->sample_nb, b->inlier, b->neighbor, b->gt_remove,
                           b->gt_add, rows ? b->rows_in : nullptr, rows ? b->rows_nb : nullptr, tile_lists, max_points, stream)))
        return rc;
    if ((rc = lrg_forward_rows(weights, b->inlier, b->neighbor, n_slots, params->n_inlier, params->n_neighbor, rows ? b->rows_in : nullptr,
                               rows ? b->rows_nb : nullptr, b->add_logits, b->rmv_logits, b->workspace, b->workspace_bytes, forward_flags,
                               stream)))
        return rc;
    if ((rc = lrg_mask_update(slots, rooms, n_slots, params, b->inlier, b->neighbor, b->center, b->add_logits, b->rmv_logits, b->gt_remove,
                              b->gt_add, nullptr, nullptr, rows ? b->sample_in : nullptr, rows ? b->sample_nb : nullptr, b->stats, stream)))
        return rc;
    return lrg_bbox_stop(slots, rooms, n_slots, max_points, params, stream);
}

// `iterations` lock-step iterations captured once into a HIP graph: every kernel argument of lrg_grow_step_packed is a device
// pointer or a constant, so a replay is the same work with one host call instead of 4-6 launches per iteration.
struct LrgStepGraph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    int iterations;
};

int lrg_step_graph_create(LrgSlot *slots, LrgRoom *rooms, int n_slots, int max_points, const LrgGrowParams *params,
                          const LrgWeights *weights, const LrgPackedBuffers *buffers, int iterations, void *stream, void **graph_out) {
    if (!graph_out || iterations < 1 || !stream) return LRG_EINVAL - 1;      // (the null stream cannot be captured)
    if (!weights || !weights->packed) return LRG_EINVAL - 2;                 // re-packing the weights per step has no place in a graph
    hipStream_t st = (hipStream_t)stream;
    LRG_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    for (int i = 0; i < iterations && rc == 0; ++i)
        rc = lrg_grow_step_packed(slots, rooms, n_slots, max_points, params, weights, buffers, stream);
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return -(int)e;
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(graph); return -(int)e; }
    LrgStepGraph *g = new LrgStepGraph{graph, exec, iterations};
    *graph_out = g;
    return 0;
}

int lrg_step_graph_launch(void *graph, void *stream) {
    if (!graph) return LRG_EINVAL - 1;
    LRG_HIP_CHECK(hipGraphLaunch(static_cast<LrgStepGraph *>(graph)->exec, (hipStream_t)stream));
    return 0;
}

int lrg_step_graph_destroy(void *graph) {
    if (!graph) return 0;
    LrgStepGraph *g = static_cast<LrgStepGraph *>(graph);
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return 0;
}

int lrg_stream_create_cu_mask(const uint32_t *mask, int words, void **stream) {
    if (!stream || (mask && words <= 0)) return LRG_EINVAL - 1;
    hipStream_t st = nullptr;
    hipError_t e = mask ? hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask) : hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) return -(int)e;
    *stream = st;
    return 0;
}

int lrg_stream_destroy(void *stream) {
    if (!stream) return 0;
    hipError_t e = hipStreamDestroy(static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : -(int)e;
}

// workspace of one room: [64 bytes: counts of a batch] | list [n] | best [n] (64-bit); a batch: the counts once, then list | best per room
static size_t nn1_room_bytes(int n) { return lrg_align_up((size_t)n * 4, 256) + lrg_align_up((size_t)n * 8, 256); }
size_t lrg_nn1_fill_workspace_bytes(int n) { return n <= 0 ? 0 : 256 + nn1_room_bytes(n); }
size_t lrg_nn1_fill_batch_workspace_bytes(const LrgFillJob *jobs, int n_jobs) {
    if (!jobs || n_jobs <= 0) return 0;
    size_t worst = 0;
    for (int g = 0; g < n_jobs; g += LRG_FILL_BATCH) {      // (groups of LRG_FILL_BATCH rooms run one after the other in the same workspace)
        size_t b = 256;
        for (int j = g; j < min(n_jobs, g + LRG_FILL_BATCH); ++j) b += jobs[j].n > 0 ? nn1_room_bytes(jobs[j].n) : 0;
        worst = max(worst, b);
    }
    return worst;
}

int lrg_nn1_fill_batch(const LrgFillJob *jobs, int n_jobs, int F, void *workspace, size_t workspace_bytes, void *stream) {
    if (n_jobs == 0) return 0;
    if (!jobs || n_jobs < 0 || F < 1 || F > 13) return LRG_EINVAL - 1;
    for (int j = 0; j < n_jobs; ++j)
        if (jobs[j].n < 0 || (jobs[j].n > 0 && (!jobs[j].points || !jobs[j].label_in || !jobs[j].label_out))) return LRG_EINVAL - 1;
    if (!workspace || workspace_bytes < lrg_nn1_fill_batch_workspace_bytes(jobs, n_jobs) || ((uintptr_t)workspace & 255)) return LRG_EINVAL - 2;
    hipStream_t st = (hipStream_t)stream;
    static const bool generic = getenv("LRG_NN1_GENERIC") != nullptr;      // (A/B switch)
    for (int g = 0; g < n_jobs; g += LRG_FILL_BATCH) {
        LrgFillBatchArgs B = {};
        B.counts = static_cast<int32_t *>(workspace);
        char *w = static_cast<char *>(workspace) + 256;
        int nb = 0, nmax = 0;
        for (int j = g; j < min(n_jobs, g + LRG_FILL_BATCH); ++j) {
            const int n = jobs[j].n;
            if (n == 0) continue;
            B.points[nb] = jobs[j].points; B.label_in[nb] = jobs[j].label_in; B.label_out[nb] = jobs[j].label_out; B.n[nb] = n;
            B.list[nb] = reinterpret_cast<int32_t *>(w);
            B.best[nb] = reinterpret_cast<unsigned long long *>(w + lrg_align_up((size_t)n * 4, 256));
            w += nn1_room_bytes(n);
            nmax = max(nmax, n);
            ++nb;
        }
        if (!nb) continue;
        LRG_HIP_CHECK(hipMemsetAsync(B.counts, 0, LRG_FILL_BATCH * sizeof(int32_t), st));
        static_assert(LRG_FILL_BATCH * sizeof(int32_t) <= 256 && sizeof(LrgFillBatchArgs) <= 4096, "counters in the workspace's first 256 bytes, arguments in the kernarg segment");
        hipLaunchKernelGGL(lrg_nn1_prep_kernel, dim3((nmax + 255) / 256, 1, nb), dim3(256), 0, st, B);
        // Query-block columns of the search grid: every column of a chunk's row stages that chunk again, and a workgroup's rounds (64 queries each) share its staging --
        // as few columns as still fill the chip (16 columns whatever the batch: 0.18 / 0.21 of the fp32 vector peak on the Area-5 set / 100 k-point scenes, 2: 0.23 / 0.27,
        // profiles/r04_fill_grid_ab.txt).  A lone small room keeps its 16.
        static const int gx_env = getenv("LRG_NN1_GX") ? atoi(getenv("LRG_NN1_GX")) : 0;            // (A/B switches)
        static const int wgs_env = getenv("LRG_NN1_WGS") ? atoi(getenv("LRG_NN1_WGS")) : 0;
        long chunks = 0;
        for (int k = 0; k < nb; ++k) chunks += (B.n[k] + LRG_NN1_C - 1) / LRG_NN1_C;
        const int want = (int)(((wgs_env > 0 ? wgs_env : LRG_NN1_TARGET_WGS) + chunks - 1) / chunks);
        const dim3 grid(min(gx_env > 0 ? gx_env : max(1, min(16, want)), (nmax + LRG_NN1_Q - 1) / LRG_NN1_Q), (nmax + LRG_NN1_C - 1) / LRG_NN1_C, nb);
        // the feature counts of the reference's variants (test_region_grow.py:72-77) are compiled in; any other goes the generic way
        if (F == 13 && !generic) hipLaunchKernelGGL(lrg_nn1_search_pairs_kernel<13>, grid, dim3(256), 0, st, B);
        else if (F == 12 && !generic) hipLaunchKernelGGL(lrg_nn1_search_pairs_kernel<12>, grid, dim3(256), 0, st, B);
        else if (F == 9 && !generic) hipLaunchKernelGGL(lrg_nn1_search_pairs_kernel<9>, grid, dim3(256), 0, st, B);
        else if (F == 6 && !generic) hipLaunchKernelGGL(lrg_nn1_search_pairs_kernel<6>, grid, dim3(256), 0, st, B);
        else hipLaunchKernelGGL(lrg_nn1_search_kernel, grid, dim3(256), 0, st, B, F);
        hipLaunchKernelGGL(lrg_nn1_write_kernel, dim3((nmax + 255) / 256, 1, nb), dim3(256), 0, st, B);
        LRG_LAUNCH_CHECK();
    }
    return 0;
}

int lrg_nn1_fill_ws(const float *points, int n, int F, const int32_t *label_in, int32_t *label_out, void *workspace,
                    size_t workspace_bytes, void *stream) {
    if (!points || !label_in || !label_out || n < 0 || F < 1 || F > 13) return LRG_EINVAL - 1;
    if (n == 0) return 0;
    LrgFillJob job = {points, label_in, label_out, n, 0};
    return lrg_nn1_fill_batch(&job, 1, F, workspace, workspace_bytes, stream);
}

int lrg_nn1_fill(const float *points, int n, int F, const int32_t *label_in, int32_t *label_out, void *stream) {
    if (!points || !label_in || !label_out || n < 0 || F < 1 || F > 16) return LRG_EINVAL - 1;
    if (n == 0) return 0;
    hipLaunchKernelGGL(lrg_nn1_fill_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, points, n, F, label_in,
                       label_out);
    LRG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
