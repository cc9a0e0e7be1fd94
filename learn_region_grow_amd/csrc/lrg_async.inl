// Free-running region growing (lrg_grow_async): ONE launch in which every slot runs its own loop of test_region_grow.py:208-306
// -- front (mask update, stop decision, commit / next seed, box query, medians, sampling, gather), the two branch stacks, the
// pooled product of the heads, the two head stacks, front again -- at its own pace.  Included by lrg_grow.hip.
//
// lrg_grow_step_packed runs the same five stages as five launches over ALL slots: every launch lasts as long as its slowest slot
// or tile (front kernel 27 us with a median slot at 14.6 us, branch launch 30 us with a lone tile at 22 us,
// profiles/r02_g_bench_kernel_stats.csv), and two thirds of the chip idle meanwhile.  Rooms are independent
// (test_region_grow.py:110-183), so nothing but those launch boundaries ties a slot with a 60-point region to the one with 4 k
// points.  Here the stages of a slot are ordered by that slot's own arrival counters:
//
//   front workgroups (1024 threads, the code of lrg_front_greedy_kernel): each serves a few slots; after the gather of a slot it
//     publishes one task per 32-row tile of the slot's rows and turns to its next slot; a slot is served again when the last of
//     its head tiles has arrived.
//   worker workgroups (the rest of the CUs, one each): teams of four wavefronts (one per SIMD) that pull tasks from one queue:
//       branch tile (slot, side, tile)   lrg_fused_tile on 32 of the slot's rows; the last tile of the slot to arrive publishes ...
//       pooled product (slot, head, 64 columns)  the arithmetic of lrg_head_gemv_kernel; the last block to arrive publishes ...
//       head tile (slot, head, tile)     lrg_fused_tile -> logits; the last one is what the slot's front workgroup waits for.
//
// Nothing waits for a workgroup that could still be waiting to be dispatched: the grid is one workgroup per CU (1024 threads each,
// accounted 128 VGPRs: a CU holds exactly one), producers never wait for consumers, and every spin is bounded by the wall clock --
// a lost hand-over raises the abort word (reported by the host as an error) instead of hanging the GPU.
//
// Hand-over between workgroups (per-XCD L2s are not coherent with each other, L1s never refreshed): payload stored write-through
// (sc1), every storing wavefront drains (`s_waitcnt vmcnt(0)`), barrier, ONE lane publishes (queue entry / arrival counter:
// agent-scope atomics); the consumer reads the payload with sc1 loads, no fences (MI355X_MICROARCH.md, inter-workgroup
// visibility, form R1; lrg_fused_tile.inl COH).  A slot's private state (masks, index lists, slot and room structs) stays with
// its front workgroup -- one CU for the whole launch -- and needs none of this.
//
// Results are those of lrg_grow_step_packed bit for bit: the same front code, the same tile code on the same rows (a slot's rows
// padded to whole tiles with copies of its last row, which neither the max-pool nor anybody's logits notice), the pooled product
// in the summation order of lrg_head_gemv_kernel.

#define LRG_AQ_TAIL 0            // control words of the queue (ints), one 64-byte line each
#define LRG_AQ_HEAD 16
#define LRG_AQ_FRONTS_DONE 32
#define LRG_AQ_ABORT 48
#define LRG_AQ_RING 64
#define LRG_ASYNC_SYNC_WORDS 16  // per slot: 0 branch tiles done, 1 their target, 2 pooled-product blocks done, 3 target, 4 head tiles done, 5 target,
                                 //           6 inlier tiles, 7 neighbour tiles of the evaluation in flight
#define LRG_ASYNC_MAX_SERVED 8   // slots per front workgroup
#define LRG_TASK_BRANCH 1
#define LRG_TASK_GEMV 2
#define LRG_TASK_HEAD 3
#define LRG_TASK(type, slot, side, idx) (((type) << 28) | ((slot) << 8) | ((side) << 7) | (idx))

// LDS of a tile team: the head stack's tile is the larger one
#define LRG_ASYNC_TILE_FLOATS LRG_TILE_LDS_FLOATS(32 * 260, 32 * 68, 1, true)
#define LRG_ASYNC_TEAM_FLOATS (LRG_ASYNC_TILE_FLOATS + 24)      // + task word, barrier counter (16-byte multiples)

struct LrgAsyncArgs {
    LrgFusedProb prob[4];        // 0 inlier branch, 1 neighbour branch, 2 add head (neighbour rows), 3 remove head (inlier rows)
    LrgGemvArgs gemv;
    LrgFrontArgs front;
    int32_t *queue;              // control words + ring
    int32_t *sync;               // [n_slots, LRG_ASYNC_SYNC_WORDS]
    int32_t *big;
    int qmask;                   // ring entries - 1 (power of two)
    int n_slots, n_front, teams;
    int max_steps;               // evaluations per slot in this launch
    long long budget_ticks;      // wall_clock64 ticks (100 MHz) after which no new evaluation is started
    long long abort_ticks;       // ... after which a waiting workgroup gives up
};

__device__ __forceinline__ void lrg_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- publishing `n` tasks: one reservation, then the entries (lanes 0 .. n-1 of the calling wavefront; n <= 64) ----
template <class F>
__device__ __forceinline__ void lrg_async_push(const LrgAsyncArgs &A, int n, int lane, F code_of) {
    int base = 0;
    if (lane == 0) base = __hip_atomic_fetch_add(&A.queue[LRG_AQ_TAIL], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = __shfl(base, 0);
    if (lane < n) lrg_st_coh(&A.queue[LRG_AQ_RING + ((base + lane) & A.qmask)], code_of(lane));
}

// ---- pooled product of a head's first layer for ONE slot and 64 columns (lrg_head_gemv_kernel's arithmetic: eight K ranges summed
//      one after the other, then their partial sums in order, then the bias) by a team of four wavefronts ----
template <class TEAM>
__device__ __forceinline__ void lrg_async_gemv(const LrgGemvArgs &g, int slot, int z, int cb, float *sm, const TEAM &team) {
    const int tid = team.tid(), lane = tid & 63, wave = tid >> 6;
    float *pl = sm, *part = sm + g.P;          // [P] pooled row, [8][64] partial sums
    for (int i = 2 * tid; i < g.P; i += 2 * FTHREADS) {
        const float2 v = lrg_ld_coh2(g.pooled + (long)slot * g.P + i);
        pl[i] = v.x; pl[i + 1] = v.y;
    }
    team.sync();
    const int c = cb * 64 + lane;
    const int kq = (g.P + 7) / 8;
    if (c < g.C) {
        const float *w = g.w[z] + c;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = wave + 4 * h;
            const int k0 = r * kq, k1 = min(g.P, k0 + kq);
            float acc = 0.f;
#pragma unroll 16
            for (int k = k0; k < k1; ++k) acc = fmaf(pl[k], w[(long)k * g.ldw], acc);
            part[r * 64 + lane] = acc;
        }
    }
    team.sync();
    if (wave == 0 && c < g.C) {
        float s = part[lane];
#pragma unroll
        for (int r = 1; r < 8; ++r) s += part[r * 64 + lane];
        lrg_st_coh(g.hb[z] + (long)slot * g.C + c, s + (g.bias[z] ? g.bias[z][c] : 0.f));
    }
}

// ---- a worker team: tasks until every front workgroup is done ----
__device__ __forceinline__ void lrg_async_worker(const LrgAsyncArgs &A, float *sm, const LrgLdsTeam &team, long long t_launch) {
    const int tid = team.tid(), lane = tid & 63;
    int *word = reinterpret_cast<int *>(sm + LRG_ASYNC_TILE_FLOATS);       // [0] task of this round
    const int row_stride = A.front.row_stride;
    const int n_gemv_blocks = (A.gemv.C + 63) / 64;
    for (;;) {
        if (tid == 0) {
            const int t = __hip_atomic_fetch_add(&A.queue[LRG_AQ_HEAD], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int *slot = &A.queue[LRG_AQ_RING + (t & A.qmask)];
            int code = 0;
            for (unsigned spin = 0;; ++spin) {
                code = lrg_ld_coh(slot);
                if (code) break;
                if ((spin & 7) == 7) {
                    if (lrg_ld_coh(&A.queue[LRG_AQ_FRONTS_DONE]) >= A.n_front || lrg_ld_coh(&A.queue[LRG_AQ_ABORT])) { code = -1; break; }
                    if ((spin & 1023) == 1023 && wall_clock64() - t_launch > A.abort_ticks) {
                        lrg_st_coh(&A.queue[LRG_AQ_ABORT], 2);
                        code = -1;
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (code > 0) lrg_st_coh(slot, 0);
            word[0] = code;
        }
        team.sync();
        const int code = word[0];
        team.sync();                                         // (read by everybody before thread 0 writes the next one)
        if (code < 0) return;
        const int type = (code >> 28) & 7, slot = (code >> 8) & 0xFFFFF, side = (code >> 7) & 1, idx = code & 127;
        int32_t *sy = A.sync + (long)slot * LRG_ASYNC_SYNC_WORDS;
        if (type == LRG_TASK_BRANCH) {
            const LrgFusedProb &P = A.prob[side];
            const long r0 = (long)slot * row_stride + (long)idx * 32;
            lrg_fused_tile<32 * 68, 32 * 132, 1, 4, false, true, true>(P, r0, 0, idx, 0x7fffffff, 0x7fffffff, sm, team, nullptr);
            lrg_drain_stores();                              // conv[1] rows and the pooled maxima are out before the arrival
            team.sync();
            if (tid < 64) {
                int last = 0;
                if (lane == 0) {
                    const int done = __hip_atomic_fetch_add(&sy[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                    last = done == lrg_ld_coh(&sy[1]);
                }
                if (__shfl(last, 0))                         // the slot's pooled feature is complete: its product with the heads' first layers
                    lrg_async_push(A, 2 * n_gemv_blocks, lane, [&](int i) { return LRG_TASK(LRG_TASK_GEMV, slot, i / n_gemv_blocks, i % n_gemv_blocks); });
            }
        } else if (type == LRG_TASK_GEMV) {
            lrg_async_gemv(A.gemv, slot, side, idx, sm, team);
            lrg_drain_stores();
            team.sync();
            if (tid < 64) {
                int last = 0, nt_in = 0, nt_nb = 0;
                if (lane == 0) {
                    const int done = __hip_atomic_fetch_add(&sy[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                    last = done == lrg_ld_coh(&sy[3]);
                    if (last) { nt_in = lrg_ld_coh(&sy[6]); nt_nb = lrg_ld_coh(&sy[7]); }
                }
                if (__shfl(last, 0)) {                       // head 0 = add on the neighbour rows, head 1 = remove on the inlier rows
                    nt_in = __shfl(nt_in, 0); nt_nb = __shfl(nt_nb, 0);
                    lrg_async_push(A, nt_nb + nt_in, lane, [&](int i) {
                        return i < nt_nb ? LRG_TASK(LRG_TASK_HEAD, slot, 0, i) : LRG_TASK(LRG_TASK_HEAD, slot, 1, i - nt_nb);
                    });
                }
            }
        } else {
            const LrgFusedProb &P = A.prob[2 + side];
            const long r0 = (long)slot * row_stride + (long)idx * 32;
            lrg_fused_tile<32 * 260, 32 * 68, 1, 4, false, true, true>(P, r0, 0, idx, 0x7fffffff, 0x7fffffff, sm, team, nullptr);
            lrg_drain_stores();                              // the logits are out before the arrival the front workgroup polls
            team.sync();
            if (tid == 0) __hip_atomic_fetch_add(&sy[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct LrgAsyncFrontCtl {
    int state[LRG_ASYNC_MAX_SERVED];     // 0 to be served, 1 evaluation in flight, 2 finished for this launch
    int steps[LRG_ASYNC_MAX_SERVED];
    int tgt[LRG_ASYNC_MAX_SERVED][3];    // running targets of the slot's three arrival counters
    int bc[4];                           // broadcasts of thread 0
};

__global__ __launch_bounds__(LRG_FRONT_THREADS) void lrg_grow_async_kernel(LrgSlot *slots, LrgRoom *rooms, LrgGrowParams prm, LrgAsyncArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const long long t_launch = wall_clock64();
    if ((int)blockIdx.x >= A.n_front) {
        // ---------------------------------------------- worker workgroup ----------------------------------------------
        const int t = tid >> 8;                              // team = four consecutive wavefronts (one per SIMD)
        if (t >= A.teams) return;
        float *sm = smem + (long)t * LRG_ASYNC_TEAM_FLOATS;
        int *word = reinterpret_cast<int *>(sm + LRG_ASYNC_TILE_FLOATS);
        if ((tid & 255) == 0) word[4] = 0;                   // the team's barrier counter (its first sync() follows thread 0's own LDS store)
        LrgLdsTeam team;
        team.cnt = &word[4];
        team.target = 0;
        team.base = t * 256;
        team.gave_up = &A.queue[LRG_AQ_ABORT];
        team.deadline = t_launch + A.abort_ticks + 100000000LL;      // (a second after everybody else has given up)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (a wavefront of the team that runs ahead adds to the counter only after thread 0's wavefront zeroed it if it waits for that
        //  store: wavefronts of one workgroup start together, but not in lock step -- so the first meeting is a plain one)
        __builtin_amdgcn_s_barrier();                        // the only workgroup-wide barrier of a worker: before any team has started
        lrg_async_worker(A, sm, team, t_launch);
        return;
    }
    // ------------------------------------------------- front workgroup -------------------------------------------------
    LrgFrontShared &SH = *reinterpret_cast<LrgFrontShared *>(smem);
    LrgAsyncFrontCtl &C = *reinterpret_cast<LrgAsyncFrontCtl *>(reinterpret_cast<char *>(smem) + ((sizeof(LrgFrontShared) + 15) & ~(size_t)15));
    const int f = blockIdx.x;
    const int n_served = (A.n_slots - f + A.n_front - 1) / A.n_front;          // slots f, f + n_front, ...
    const int row_stride = A.front.row_stride;
    const int n_gemv = 2 * ((A.gemv.C + 63) / 64);
    if (tid < LRG_ASYNC_MAX_SERVED) { C.state[tid] = tid < n_served ? 0 : 2; C.steps[tid] = 0; C.tgt[tid][0] = C.tgt[tid][1] = C.tgt[tid][2] = 0; }
    // a slot's rows have a fixed place in the row arrays: their tags are written once per launch
    for (int i = 0; i < n_served; ++i) {
        const int s = f + i * A.n_front;
        for (int j = tid; j < row_stride; j += LRG_FRONT_THREADS) {
            lrg_st_coh(&A.front.row_slot_in[(long)s * row_stride + j], s);
            lrg_st_coh(&A.front.row_slot_nb[(long)s * row_stride + j], s);
        }
    }
    lrg_drain_stores();
    __syncthreads();
    for (;;) {
        // a hand-over given up anywhere (or this launch far beyond any sane duration): everybody leaves, the host reports it
        if (tid == 0) {
            int ab = lrg_ld_coh(&A.queue[LRG_AQ_ABORT]);
            if (!ab && wall_clock64() - t_launch > A.abort_ticks) { ab = 1; lrg_st_coh(&A.queue[LRG_AQ_ABORT], 1); }
            C.bc[2] = ab;
        }
        __syncthreads();
        const int aborted = C.bc[2];
        __syncthreads();
        if (aborted) {
            if (tid == 0 && A.front.stats) atomicAdd(reinterpret_cast<unsigned long long *>(&A.front.stats[3]), 1ULL);
            break;
        }
        int live = 0;
        for (int i = 0; i < n_served; ++i) {
            const int s = f + i * A.n_front;
            int st = C.state[i];
            if (st == 2) continue;
            ++live;
            if (st == 1) {
                if (tid == 0) C.bc[0] = lrg_ld_coh(&A.sync[(long)s * LRG_ASYNC_SYNC_WORDS + 4]) >= C.tgt[i][2];
                __syncthreads();
                const int ready = C.bc[0];
                __syncthreads();
                if (!ready) continue;
                st = 0;
            }
            // a new evaluation only within the budget of this launch
            if (tid == 0) {
                const long long el = wall_clock64() - t_launch;
                C.bc[1] = C.steps[i] >= A.max_steps || el > A.budget_ticks;
            }
            __syncthreads();
            const int stop = C.bc[1];
            __syncthreads();
            if (stop) { if (tid == 0) C.state[i] = 2; continue; }
            const int r = lrg_front_greedy_slot<true>(SH, slots, rooms, A.n_slots, prm, A.front, A.big, s);
            if (r == 0) {
                // no evaluation: the slot is idle / its room finished (-> finished for this launch), or it stopped a region / goes on
                // looking for a seed (-> served again at once)
                __syncthreads();
                if (tid == 0) {
                    const int status = slots[s].status;
                    C.state[i] = (slots[s].room < 0 || status == LRG_DONE || status == LRG_IDLE) ? 2 : 0;
                }
                __syncthreads();
                continue;
            }
            // rows, centre and the zeroed pooled feature are out (write-through) once every wavefront has drained
            lrg_drain_stores();
            __syncthreads();
            const int nt_in = ((r >> 16) + 31) >> 5, nt_nb = ((r & 0xFFFF) + 31) >> 5;
            if (tid < 64) {
                if (lane == 0) {
                    int32_t *sy = A.sync + (long)s * LRG_ASYNC_SYNC_WORDS;
                    C.tgt[i][0] += nt_in + nt_nb; C.tgt[i][1] += n_gemv; C.tgt[i][2] += nt_in + nt_nb;
                    lrg_st_coh(&sy[1], C.tgt[i][0]); lrg_st_coh(&sy[3], C.tgt[i][1]); lrg_st_coh(&sy[5], C.tgt[i][2]);
                    lrg_st_coh(&sy[6], nt_in); lrg_st_coh(&sy[7], nt_nb);
                    C.state[i] = 1;
                    C.steps[i] += 1;
                }
                lrg_drain_stores();
                lrg_async_push(A, nt_in + nt_nb, lane, [&](int k) {
                    return k < nt_in ? LRG_TASK(LRG_TASK_BRANCH, s, 0, k) : LRG_TASK(LRG_TASK_BRANCH, s, 1, k - nt_in);
                });
            }
            __syncthreads();
        }
        if (!live) break;
        __builtin_amdgcn_s_sleep(4);
    }
    lrg_drain_stores();
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&A.queue[LRG_AQ_FRONTS_DONE], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
