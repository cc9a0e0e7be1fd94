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
//       pooled product (slot, head, 128 columns)  the arithmetic of lrg_head_gemv_kernel; the last block to arrive publishes ...
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
// in the one summation order all formulations share (lrg_head_gemv_kernel, lrg_net.hip).

#ifndef LRG_ASYNC_HEAD_PRIO
#define LRG_ASYNC_HEAD_PRIO 1       // wave priority (s_setprio) of a team while it runs a head tile
#endif
#ifndef LRG_WORKER_POLL_SLEEP
#define LRG_WORKER_POLL_SLEEP 8    // s_sleep argument (x 64 cycles) between two looks of an idle team at its ring entry (x LrgAsyncBuffers.poll_sleep)
#endif
#ifndef LRG_WAIT_POOLED_SLEEP
#define LRG_WAIT_POOLED_SLEEP 2     // s_sleep argument (x 64 cycles) between two looks of a head tile's wavefronts at their slot's pooled-product counter
#endif
#ifndef LRG_ASYNC_FD
#define LRG_ASYNC_FD 4              // depth of the tile teams' weight ring (k-groups in flight)
#endif
#define LRG_AQ_TAIL 0            // control words of the queue (ints), one 64-byte line each; ring 1 (pooled blocks and head tiles when
#define LRG_AQ_HEAD 16           // the workgroups run more than one team): + LRG_AQ_SECOND
#define LRG_AQ_FRONTS_DONE 32
#define LRG_AQ_ABORT 48
#define LRG_AQ_SECOND 64
#define LRG_AQ_GTAIL 96          // entries written to the pooled-product units' ring so far
#define LRG_AQ_ARRIVED 112       // workgroups of this launch that have started (the start rendezvous of the front workgroups)
#ifndef LRG_ASYNC_START_TICKS
#define LRG_ASYNC_START_TICKS 2000000LL      // 20 ms (wall_clock64: 100 MHz): by then every workgroup of the launch has started, or never will while the others wait
#endif
#define LRG_AQ_FTAIL 128         // the fill-in ring (tasks of the in-launch 1-NN fill-in, test_region_grow.py:308-316): entries reserved / taken
#define LRG_AQ_FHEAD 144
#define LRG_AQ_RING 192          // ring 0, then ring 1 (qmask + 1 entries each), then the units' ring (gmask + 1 entries), then the fill-in ring (fmask + 1)
#define LRG_ASYNC_FILL_RING 8192 // entries of the fill-in ring: one per 256 candidate points of a finished room (a 131 072-point scene: 512)
// behind the fill-in ring: the wave rings' control words (ring t = side * 4 + quarter: [32 t] entries reserved, [32 t + 16] tickets taken) and the eight rings
#define LRG_AQ_WAVE(A) (LRG_AQ_RING + 2 * ((A).qmask + 1) + ((A).gmask + 1) + LRG_ASYNC_FILL_RING)
#define LRG_AQ_WAVE_RING(A, t) (LRG_AQ_WAVE(A) + 256 + (t) * ((A).wmask + 1))
#define LRG_WORKER_THREADS 512   // workgroup of lrg_grow_async_worker_kernel
#define LRG_TASK_FILL 4
#define LRG_ASYNC_SYNC_WORDS 16  // per slot: 0 branch tiles done, 1 pooled-product blocks done, 2 head tiles done (arrival counters); 4 .. 7 one 16-byte word written by
                                 //           the front workgroup: the three targets and inlier | neighbour << 16 tiles of the evaluation in flight; 8 a debug stamp
#define LRG_ASYNC_MAX_SERVED 16   // slots per front workgroup
#define LRG_TASK_BRANCH 1
#define LRG_TASK_GEMV 2
#define LRG_TASK_HEAD 3
#define LRG_TASK(type, slot, side, idx) (((type) << 28) | ((slot) << 8) | ((side) << 7) | (idx))

// LDS of a tile team: [task word, barrier counter (16-byte multiples) (+ the cycle stamps of an LRG_TRACE build)][the tile's buffers].  The head stack's tile is the
// larger one; a team that only ever runs branch tiles (the first `small_teams` of a workgroup, where four teams share the CU's 160 KB) gets by with the smaller.
#define LRG_ASYNC_TILE_FLOATS LRG_TILE_LDS_FLOATS(32 * 260, 32 * 68, 1, true)
#define LRG_ASYNC_BRANCH_TILE_FLOATS LRG_TILE_LDS_FLOATS(32 * 68, 32 * 132, 1, true)
#define LRG_ASYNC_CTL_FLOATS (24 + (LRG_TRACE ? 64 : 0))
#define LRG_ASYNC_TEAM_FLOATS (LRG_ASYNC_TILE_FLOATS + LRG_ASYNC_CTL_FLOATS)
#define LRG_ASYNC_SMALL_TEAM_FLOATS (LRG_ASYNC_BRANCH_TILE_FLOATS + LRG_ASYNC_CTL_FLOATS)
// ... and a team that only serves the fill-in ring (the EXTRA team of a fill workgroup, always its last) with a chunk's rows, label flags and partial minima
// (lrg_async_fill_chunk: 256 x 13 + 256 + 4 x 64 x 2 floats): three tile teams and it are 150 KB
#define LRG_ASYNC_FILL_TEAM_FLOATS (LRG_NN1_C * 13 + LRG_NN1_C + 4 * 64 * 2 + 8 + LRG_ASYNC_CTL_FLOATS)

struct LrgAsyncArgs {
    LrgFusedProb prob[4];        // 0 inlier branch, 1 neighbour branch, 2 add head (neighbour rows), 3 remove head (inlier rows)
    LrgGemvArgs gemv;
    LrgFrontArgs front;
    int32_t *queue;              // control words + ring
    int32_t *sync;               // [n_slots, LRG_ASYNC_SYNC_WORDS]
    int32_t *big;
    int32_t *room_queue;         // nullable: [0] rooms handed out so far, [1] rooms queued, [2 + k] = room index | reset << 30
    int qmask;                   // ring entries - 1 (power of two)
    int gmask;                   // entries of the pooled-product units' ring - 1 (power of two, at least 2 n_slots)
    int gemv_batch;              // > 1 (without the units): pooled products in batches of up to so many slots (LRG_GEMV_BATCH) -- the slots whose branch tiles are all in
                                 // queue up in the (otherwise unused) units' ring; a batch's blocks stream the kernels' 128 columns ONCE for all its slots
    long long gemv_batch_ticks;  // a batch that is not full that long after its leader task was taken is closed with the slots it has
    int gemv_units;              // workgroups n_front .. n_front + gemv_units - 1 hold 32 columns each of the heads' pooled kernels in LDS (0: the
                                 // pooled product is a task of the tile teams, 128 columns each)
    int n_slots, n_front, teams;
    int head_ring;               // the ring pooled blocks and head tiles are published to: 1, or 0 = one ring for all tasks and all teams
    int ring0_halves;            // more than one team per workgroup: team t of worker workgroup w runs branch tiles (ring 0) if 2 t + (w & 1) < ring0_halves,
                                 // else pooled blocks and head tiles (ring 1) -- 2: the first team everywhere, 3: one and a half teams on average, ...
    // in-launch fill-in (nullable: fill_list == nullptr -> the host fills finished rooms in between launches)
    int32_t *fill_list;          // [points of all rooms] per room (at the room's offset in the arenas): indices of its unlabeled points
    unsigned long long *fill_best;   // [points of all rooms] best (distance bits << 32 | index) per point
    int32_t *fill_sync;          // [n_rooms, 4]: unlabeled points, candidate chunks done, chunks in all, filled
    const int32_t *fill_label_base;  // the label arena (LrgRoom.label points into it) and the filled-label arena of the same layout
    int32_t *fill_out_base;
    int fill_wgs;                // the last team of the first fill_wgs worker workgroups serves the fill-in ring only
    int small_teams;             // the first so many teams of a worker workgroup run branch tiles only, on the smaller LDS region (four teams per workgroup)
    int small_alt;               // 1: ... and one more of them on the odd workgroups
    int fill_extra;              // 1: ... and that team is one more than the other workgroups have (where LDS and threads allow: up to three tile teams)
    int fill_hybrid;             // 1: four tile teams per CU, the fill-in team is one of them: it takes a fill-in task when one is waiting and ring 1's next task otherwise
    // Shared tail tiles (nullable: tail == nullptr -> every slot pads its own last tile).  A slot's rows beyond its last full 32-row tile -- 16 of 91 rows per side
    // on average: 18 % of all tile rows were such padding -- are reserved from a cursor per side in rows that all slots share (LrgFrontArgs.tail_*), so that the
    // tails of several slots fill one BRANCH tile: the tile code's packed form (runs of rows of one slot each: per-run max-pool, lrg_forward_packed's arithmetic bit
    // for bit).  A tile is published by whoever brings its count of written rows to 32; a slot whose last tile stays open longer than tail_ticks closes it (the
    // cursor is moved to the tile's end, the missing rows count as dead).  The HEAD stack of a tail stays a tile of the slot's own: it reads the slot's conv[1] rows
    // where the shared tile left them and stores the logits of the slot's rows only (lrg_fused_tile: nrows_out).
    int32_t *tail;               // [0] / [16] the sides' row cursors (= LrgFrontArgs.tail_cur); [32 + side * tail_tiles + tile] rows accounted for | dead rows << 16
    int tail_tiles;              // shared tiles per side
    int tail_heads;              // 1 (without the units): the HEAD stacks of the tails run on the shared tiles too -- a shared tile's head task is published when the
                                 // pooled products of ALL slots with rows in it are complete ([32 + 2 * tail_tiles + side * tail_tiles + tile]: slots ready | the
                                 // tile's slots << 16); 0: a head tile of the slot's own per tail, storing its rows only
    long long tail_ticks;        // (wall_clock64: 100 MHz)
    float *pool_rows;            // nullable (with the units): [n_slots][2 sides][16 tiles][P / 2] column maxima by branch tile, instead of atomicMax on the pooled feature
    int pool_rows_stride;        // 2 * 16 * (P / 2)
    int poll_sleep;              // s_sleep(8) repeats between two polls of an idle team (1 = ~0.25 us)
    int branch_parts;            // tasks per branch tile (1, 2, 4): they share the column blocks of the pooled layer (lrg_fused_tile)
    // Wave-branch mode (round 6; lrg_wave_tile.inl): the launch is TWO kernels resident together -- lrg_grow_async_kernel with the front workgroups and the
    // pooled-product units only, and lrg_grow_async_worker_kernel (512 threads, up to 256 VGPRs) with `wave_wgs` wave-branch CUs and the head teams' CUs behind them.
    // A branch tile is a PREFIX task (layers 0 - 3, by one wavefront of a CU that holds those kernels of both branches in LDS) that publishes the tile's POOL tasks
    // (a quarter of the pooled layer each -- or half a quarter: wave_split 4 / 8 -- by one wavefront of a CU that holds its (side, half) of that kernel in LDS).
    // Rings of their own: 0 .. 3 = POOL tasks of (side, half), 4 = PREFIX tasks.  0: off -- one kernel, branch tiles by the tile teams.
    int wave_wgs;                // wave-branch CUs: workgroups 0 .. wave_a_wgs - 1 of the worker kernel run PREFIX tasks, wave_a_wgs .. wave_wgs - 1 POOL tasks of
    int wave_a_wgs;              //   (side, half) = (w - wave_a_wgs) & 3
    int wave_waves;              // wavefronts per wave-branch CU that run branch tasks (4: one per SIMD)
    int wave_split;              // POOL tasks per tile: 4 (a quarter = two pairs of column blocks each) or 8 (one pair each)
    int wave_fill;               // 1: wavefronts 4 .. 7 of the first fill_wgs wave-branch CUs are a fill-in team (VALU work beside the MFMA-bound branch waves)
    int wmask;                   // entries of one wave ring - 1 (power of two)
    float *h3[2];                // [row_cap, 128] per side: layer 3's output rows, from the PREFIX to the POOL tasks
    int rt_bb_every;             // register tiles: every so-manyth worker CU runs branch tiles on BOTH its teams (0: none)
    int unit_pairs;              // 1: the pooled-product units run their tasks on half-teams of two wavefronts (lrg_async_gemv_unit2)
    int reg_tiles;               // 1: the worker kernel's workgroups are all alike -- team 0 runs the branch tiles of ring 0 as REGISTER TILES (lrg_team_branch_tile_reg: a team
                                 // of four wavefronts per tile, layers 0 - 2 per wavefront in registers, one barrier), team 1 the pooled blocks and head tiles of ring 1
    int worker_base;             // blockIdx.x of the first worker workgroup in the kernel that runs the tile teams (n_front + gemv_units, or wave_wgs in the worker kernel)
    int total_wgs;               // workgroups of the launch in all (both kernels): what the start rendezvous waits for
    int max_steps;               // evaluations per slot in this launch
    long long start_ticks;       // ... the front workgroups wait at most this long for all workgroups of the launch to have started (reason 6)
    long long budget_ticks;      // wall_clock64 ticks (100 MHz) after which no new evaluation is started
    long long abort_ticks;       // ... after which a waiting workgroup gives up
    unsigned long long *work;    // nullable: [4] evaluations, distinct inlier rows, distinct neighbour rows, 32-row tiles (x 2 stacks) of this buffer's launches
    unsigned long long *dbg;     // nullable: [32] accumulators of wall-clock ticks (10 ns) for tools/free_run_perf.py --
                                 // 0 front busy, 1 front steps; per evaluation, since its tasks were published: 2 last branch tile in,
                                 // 3 last pooled-product block in, 4 last head tile in, 5 seen by the front workgroup, 6 evaluations;
                                 // 8 + 2 t busy ticks of task type t, 9 + 2 t their number; 16 ticks teams waited for a task, 17 waits
};
#ifndef LRG_ASYNC_DEBUG
#define LRG_ASYNC_DEBUG 0        // 1: the tick accumulators of LrgAsyncBuffers.debug_ticks are compiled in (tools/free_run_perf.py builds with it);
#endif                           // off by default: the stamps keep 64-bit values alive through the front and cost it registers
#define LRG_DBG(A) (LRG_ASYNC_DEBUG && (A).dbg)
__device__ __forceinline__ void lrg_dbg_add(const LrgAsyncArgs &A, int i, long long v) {
    if (LRG_DBG(A)) atomicAdd(&A.dbg[i], (unsigned long long)v);
}

#ifdef LRG_EXP_NO_DRAIN      // (experiment switch, --policy gt only: arrivals without waiting for the stores before them -- results may be stale)
__device__ __forceinline__ void lrg_drain_stores() {}
#else
__device__ __forceinline__ void lrg_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// ---- publishing `n` tasks: one reservation, then the entries (lanes 0 .. n-1 of the calling wavefront; n <= 64) ----
template <class F>
__device__ __forceinline__ void lrg_async_push(const LrgAsyncArgs &A, int ring, int n, int lane, F code_of) {
    int base = 0;
    if (lane == 0) base = __hip_atomic_fetch_add(&A.queue[LRG_AQ_TAIL + ring * LRG_AQ_SECOND], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = __shfl(base, 0);
    if (lane < n) lrg_st_coh(&A.queue[LRG_AQ_RING + ring * (A.qmask + 1) + ((base + lane) & A.qmask)], code_of(lane));
}

// the head tiles of a slot whose pooled feature is complete: head 0 = add on the neighbour rows, head 1 = remove on the inlier rows; w_in / w_nb = the side's tiles
// of the slot's own | bit 12: its tail lies in the shared rows and gets a head tile of its own there (index 16)
__device__ __forceinline__ int32_t *lrg_tail_head_word(const LrgAsyncArgs &A, int side, int tile) { return A.tail + 32 + 2 * (long)A.tail_tiles + ((long)side * A.tail_tiles + tile); }
__device__ __forceinline__ void lrg_async_push_heads(const LrgAsyncArgs &A, int slot, int w_in, int w_nb, int lane) {
    const int nt_in = w_in & 0xFFF, nt_nb = w_nb & 0xFFF;
    int sh_in = (w_in >> 12) & 1, sh_nb = (w_nb >> 12) & 1;
    if (A.tail_heads && (sh_in | sh_nb)) {
        // the slot is ready for its heads: counted into the shared tiles its tails lie in (lanes 0 / 1: the inlier side's first / second tile, 2 / 3: the neighbour
        // side's); whoever completes a tile's count publishes its head task (head 1 = remove on the inlier rows, head 0 = add on the neighbour rows)
        const int32_t *sy = A.sync + (long)slot * LRG_ASYNC_SYNC_WORDS;
        if (lane < 4) {
            const int side = lane >> 1, second = lane & 1;
            if (side ? sh_nb : sh_in) {
                const int tb = lrg_ld_coh(&sy[9 + side]), tl = (lrg_ld_coh(&sy[11]) >> (16 * side)) & 0xFFFF;
                const int ta = tb >> 5, tz = (tb + tl - 1) >> 5;
                if (!second || tz != ta) {
                    const int tile = second ? tz : ta;
                    const int old = __hip_atomic_fetch_add(lrg_tail_head_word(A, side, tile), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((old & 0xFFFF) + 1 == (int)((unsigned)old >> 16)) {
                        const int e = __hip_atomic_fetch_add(&A.queue[LRG_AQ_TAIL + A.head_ring * LRG_AQ_SECOND], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        lrg_st_coh(&A.queue[LRG_AQ_RING + A.head_ring * (A.qmask + 1) + (e & A.qmask)], LRG_TASK(LRG_TASK_HEAD, tile, side ? 0 : 1, 17));
                        if (A.work) atomicAdd(&A.work[7], 1ULL);
                    }
                }
            }
        }
        sh_in = sh_nb = 0;
    }
    lrg_async_push(A, A.head_ring, nt_nb + nt_in + sh_nb + sh_in, lane, [&](int i) {
        if (i < nt_nb) return LRG_TASK(LRG_TASK_HEAD, slot, 0, i);
        if (i < nt_nb + nt_in) return LRG_TASK(LRG_TASK_HEAD, slot, 1, i - nt_nb);
        return (i == nt_nb + nt_in && sh_nb) ? LRG_TASK(LRG_TASK_HEAD, slot, 0, 16) : LRG_TASK(LRG_TASK_HEAD, slot, 1, 16);
    });
}

// ---- pooled product of a head's first layer for ONE slot and 128 columns by a team of four wavefronts, in the summation order of
//      lrg_head_gemv_kernel / lrg_head_gemm_kernel (eight K ranges, each a chain of FMAs over k = 8g + 0, 4, 1, 5, 2, 6, 3, 7 -- what
//      v_mfma_f32_32x32x2_f32 does with its lane halves --; then the eight partial sums in order; then the bias) ----
// What it costs is the trip of 1024 x 128 weights (512 KB) from L2, not the arithmetic: every lane owns FOUR consecutive columns
// (one 16-byte load per row of the kernel) and ONE of the eight K ranges -- 16 lanes per range and column half, the four lane
// groups of a wavefront on four ranges -- with sixteen rows in flight per lane.  (With one column per lane and the loop left to the
// compiler two loads were in flight: 34 us per block of 64 columns, profiles/r03_free3_perf.log.)
#define LRG_GEMV_TASK_COLS 128
#ifndef LRG_GEMV_BATCH
#define LRG_GEMV_BATCH 8           // slots per batch of pooled products ("batched pooled products" below)
#endif
#define LRG_GEMV_NOBODY 0xFFFFF     // a ring position of a closed batch that no slot took
template <class TEAM>
__device__ __forceinline__ void lrg_async_gemv(const LrgGemvArgs &g, int slot, int z, int cb, float *sm, const TEAM &team) {
    const int tid = team.tid(), lane = tid & 63, wave = tid >> 6;
    float *pl = sm, *part = sm + g.P;          // [P] pooled row, [8][128] partial sums
    for (int i = 2 * tid; i < g.P; i += 2 * FTHREADS) {
        const float2 v = lrg_ld_coh2(g.pooled + (long)slot * g.P + i);
        pl[i] = v.x; pl[i + 1] = v.y;
    }
    team.sync();
    const int kq = g.P >> 3;                   // (a multiple of 16: checked by lrg_grow_async)
    const int half = wave >> 1, r = 4 * (wave & 1) + (lane >> 4), cl = half * 64 + 4 * (lane & 15);
    const int c = cb * LRG_GEMV_TASK_COLS + cl;
    if (c < g.C) {
        const float *w = g.w[z] + c + (long)(r * kq) * g.ldw;
        const float *p = pl + r * kq;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int kb = 0; kb < kq; kb += 16) {
            float4 wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = *reinterpret_cast<const float4 *>(w + (long)(kb + u) * g.ldw);
#pragma unroll
            for (int uu = 0; uu < 16; ++uu) {
                const int u = (uu & 8) | ((uu & 1) << 2) | ((uu >> 1) & 3);      // k = 8g + 0, 4, 1, 5, 2, 6, 3, 7: the order of the MFMA formulation
                const float pk = p[kb + u];
                acc.x = fmaf(pk, wv[u].x, acc.x); acc.y = fmaf(pk, wv[u].y, acc.y);
                acc.z = fmaf(pk, wv[u].z, acc.z); acc.w = fmaf(pk, wv[u].w, acc.w);
            }
        }
        *reinterpret_cast<float4 *>(part + r * LRG_GEMV_TASK_COLS + cl) = acc;
    }
    team.sync();
    if (tid < LRG_GEMV_TASK_COLS && cb * LRG_GEMV_TASK_COLS + tid < g.C) {
        const int col = cb * LRG_GEMV_TASK_COLS + tid;
        float s = part[tid];
#pragma unroll
        for (int q = 1; q < 8; ++q) s += part[q * LRG_GEMV_TASK_COLS + tid];
        lrg_st_coh(g.hb[z] + (long)slot * g.C + col, s + (g.bias[z] ? g.bias[z][col] : 0.f));
    }
}


// ---- pooled-product units: the heads' pooled kernels stay in LDS ----
// As a task of the tile teams the pooled product of a slot is four trips of 512 KB from L2 (7.7 us each, 12.5 us from the last branch
// tile to the last block with the queueing, profiles/r03_free8_perf.log) in the middle of the slot's chain of latencies.  The
// kernels are the same for every slot: 2 x [P = 1024, C = 256] floats = 2 MB = sixteen CUs' LDS at 32 columns each.  A unit
// (workgroup) loads its [P, 32] slice once per launch; its four teams of four wavefronts take the slots whose branch tiles have
// all arrived from ONE ring that every unit reads (entry i: (generation of i) << 20 | slot, written once by the last branch tile
// to arrive; a unit's teams take the entries in turn -- a ticket counter in the unit's LDS, nothing to reserve in memory); a task =
// the slot's 4 KB pooled row from L2, 128 FMAs per lane from LDS, 32 sums out, one more arrival on the slot's pooled-product counter,
// which the slot's head tiles poll (LrgWaitPooled).  Summation order: that of lrg_head_gemv_kernel (eight K ranges, MFMA k order
// inside, the partial sums in order, the bias last) -- bit for bit what the tile teams' blocks give.
// (Tried: mailboxes instead of the ring -- a 64-bit word per slot, arrivals | target << 32, every branch tile adds itself, one team per
// unit watches all slots -- so that the units start ~2 us earlier, before the last tile knows it was the last: 808 k -> 795 k
// instance-steps/s at 68 slots, the second atomic per tile and the watchers' loads cost more than the earlier start gains.)
#define LRG_GEMV_UNIT_COLS 32
#define LRG_GEMV_UNIT_TEAMS 4
#define LRG_GEMV_UNIT_TEAM_FLOATS(P) ((P) + 8 * LRG_GEMV_UNIT_COLS + 16)      // pooled row, partial sums, task words + barrier counter
#define LRG_GEMV_UNIT_FLOATS(P) ((P) * LRG_GEMV_UNIT_COLS + LRG_GEMV_UNIT_TEAMS * LRG_GEMV_UNIT_TEAM_FLOATS(P))      // + n_slots claim words behind
#define LRG_GEMV_UNIT_MAX_SLOTS 2048
__device__ __forceinline__ int lrg_gemv_ring_tag(int i, int gmask) { return (((unsigned)i / (unsigned)(gmask + 1)) % 2047u) + 1; }

// ---- the launch's arguments ----
// ONE kernel parameter, so that every role below can be a function of its own (own register allocation: the tile code needs 112
// VGPRs, a 1024-thread workgroup has 128 per lane -- inlined into one kernel body, the three task types and the front spilled
// ~150 dwords per lane, some of them inside the tiles' passes) and still reads the arguments the way a kernel does: scalar loads
// from the kernarg segment, nothing passed on, nothing copied to the stack.
struct LrgAsyncKArgs {
    LrgSlot *slots;
    LrgRoom *rooms;
    LrgGrowParams prm;
    LrgAsyncArgs A;
};
// (inside a non-kernel function __builtin_amdgcn_kernarg_segment_ptr() folds to null: the kernel takes the pointer and hands it on,
//  typed as constant address space, so that the roles' loads of the arguments stay scalar loads)
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) void *lrg_kargs_ptr;
#else
typedef const void *lrg_kargs_ptr;
#endif
#define LRG_ASYNC_KARGS() (*(const LrgAsyncKArgs *)kp)
// Arguments of a non-kernel function arrive in vector registers; what is wave-uniform goes back to scalar registers first, so that
// the loads of the launch's arguments are scalar loads and the addresses derived from them scalar arithmetic.
__device__ __forceinline__ lrg_kargs_ptr lrg_uniform(lrg_kargs_ptr p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (lrg_kargs_ptr)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int lrg_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// the launch's dynamic LDS: roles get OFFSETS into it (a float * parameter would be a generic pointer, and every LDS access of a
// tile a flat instruction)
extern __shared__ __attribute__((aligned(16))) float lrg_async_smem[];
#include "lrg_wave_tile.inl"
#define LRG_ASYNC_ROLE __device__ __noinline__
// The tile tasks are inlined into the worker's loop (round 4).  As functions of their own (round 3) every call saved and restored the callee-saved registers the
// tile code uses -- 54 / 72 dwords per lane and task through scratch memory: ~0.6 MB of writes and as many reads per evaluation, which is what the write counter of
// profiles/r03_pmc_free_run.json (13.5 GB per launch) was made of, and 1.3 % of the rate.  One function = one register allocation for branch and head tiles and the
// loop's few values; the compiler keeps the MFMA loops free of spills (checked in the ISA: scratch traffic only at the worker's own entry and exit, once per launch).
// The front step stays a function: inlined into the serving loop it reloads ~170 spilled values per step.
#ifndef LRG_ASYNC_TASK
#define LRG_ASYNC_TASK __device__ __forceinline__
#endif

// branch-only teams of worker workgroup wg_no (small_alt: one more on the odd ones -- a step's branch tiles are ~64 % of its tile time, 2.5 of 4 teams)
__device__ __forceinline__ int lrg_async_small_teams(const LrgAsyncArgs &A, int wg_no) { return A.small_teams + ((wg_no & 1) ? A.small_alt : 0); }

// (experiment switches: a stage made N x 10 ns longer -- the slope of the step time over N is that stage's weight in the step, queueing included;
//  the results do not change, so these run under the normal policy: tools/r04_delay.sh)
__device__ __forceinline__ void lrg_exp_delay(int ticks) {
    if (ticks > 0) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    }
}
#ifndef LRG_TICKET_EARLY
#define LRG_TICKET_EARLY 1
#endif
#ifndef LRG_EXP_DELAY_FRONT
#define LRG_EXP_DELAY_FRONT 0
#endif
#ifndef LRG_EXP_DELAY_BRANCH
#define LRG_EXP_DELAY_BRANCH 0
#endif
#ifndef LRG_EXP_DELAY_HEAD
#define LRG_EXP_DELAY_HEAD 0
#endif

__device__ __forceinline__ LrgLdsTeam lrg_async_team(const LrgAsyncArgs &A, float *sm, int target) {      // sm: the team's part of the LDS
    int *word = reinterpret_cast<int *>(sm - LRG_ASYNC_CTL_FLOATS);
    LrgLdsTeam team;
    team.cnt = &word[4];
    team.target = target;
    team.base = (int)(threadIdx.x & ~255u);
    team.gave_up = &A.queue[LRG_AQ_ABORT];
    return team;
}

// ---- shared tail tiles ----
// One word per shared tile and side: rows accounted for (written, or dead) | dead rows << 16.  The slot whose reservation takes a tile's FIRST row publishes the
// tile's task together with its own tiles (no round trip of its own); every slot adds its rows when they are out (an atomic nobody waits for); the team that takes
// the task waits for the 32nd row -- and closes the tile (moves the side's cursor to its end: the rest are dead rows) when that takes longer than tail_ticks.
__device__ __forceinline__ int32_t *lrg_tail_word(const LrgAsyncArgs &A, int side, int tile) { return A.tail + 32 + ((long)side * A.tail_tiles + tile); }
__device__ __forceinline__ void lrg_tail_account(const LrgAsyncArgs &A, int side, int tile, int cnt, int dead) {
    __hip_atomic_fetch_add(lrg_tail_word(A, side, tile), cnt | (dead << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (result unused: no wait)
}
// by ONE thread of the team that runs shared tile (side, tile): returns its dead rows once all 32 are accounted for (-1: given up)
__device__ __forceinline__ int lrg_tail_wait(const LrgAsyncArgs &A, int side, int tile, long long t_launch) {
    const int32_t *w = lrg_tail_word(A, side, tile);
    const long long t0 = wall_clock64();
    bool closed = false;
    for (unsigned spin = 1;; ++spin) {
        const int v = lrg_ld_coh(w);
        if ((v & 0xFFFF) >= 32) return v >> 16;
        if (!closed && wall_clock64() - t0 > A.tail_ticks) {
            int c = lrg_ld_coh(&A.tail[16 * side]);
            if ((c >> 5) != tile) closed = true;      // (every row of the tile is reserved: their slots are writing them)
            else if (__hip_atomic_compare_exchange_strong(&A.tail[16 * side], &c, (tile + 1) * 32, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                const int dead = (tile + 1) * 32 - c;
                lrg_tail_account(A, side, tile, dead, dead);
                closed = true;
            }
        }
        if ((spin & 255u) == 0) {
            if (lrg_ld_coh(&A.queue[LRG_AQ_ABORT])) return -1;
            if (wall_clock64() - t_launch > A.abort_ticks) { lrg_st_coh(&A.queue[LRG_AQ_ABORT], 8); return -1; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// ---- the three task types (each returns the team's barrier count, to be handed to the next one) ----
// a branch tile that has rows of `slot` is done (wavefront 0 of the team): the last one to arrive for the slot's evaluation publishes its pooled product and head tiles
__device__ __forceinline__ void lrg_async_branch_arrive(const LrgAsyncArgs &A, int slot, int lane, long long t_task, bool count_task) {
    int32_t *sy = A.sync + (long)slot * LRG_ASYNC_SYNC_WORDS;
    int last = 0, nt_in = 0, nt_nb = 0;
    if (lane == 0) {
        // (the evaluation's targets and tile counts were written before its tasks were published: fetched beside the arrival, not after it)
        const float4 tq = lrg_ld_coh4(reinterpret_cast<const float *>(sy), 16u);      // (targets and tile counts: one 16-byte word, written as one)
        const int tgt = __float_as_int(tq.x);
        nt_in = __float_as_int(tq.w) & 0xFFFF; nt_nb = (int)((unsigned)__float_as_int(tq.w) >> 16);
        const int done = __hip_atomic_fetch_add(&sy[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        last = done == tgt;
        if (LRG_DBG(A)) {
            const long long now = wall_clock64();
            if (count_task) { lrg_dbg_add(A, 8 + 2 * LRG_TASK_BRANCH, now - t_task); lrg_dbg_add(A, 9 + 2 * LRG_TASK_BRANCH, 1); }
            if (last) lrg_dbg_add(A, 2, (int)((unsigned)now - (unsigned)lrg_ld_coh(&sy[8])));
        }
    }
    if (__shfl(last, 0)) {                       // the slot's pooled feature is complete: its product with the heads' first layers
        if (A.gemv_units) {
            // one entry for all units; and the head tiles at once -- they stage their rows and run the first pass of MFMAs while the
            // units work, and wait for the product in front of that pass's epilogue (LrgWaitPooled)
            if (lane == 0) {
                // (entry: generation tag | tiles per side - 1, four bits each | slot -- the units take the maximum over the slot's tile rows)
                const int i = __hip_atomic_fetch_add(&A.queue[LRG_AQ_GTAIL], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lrg_st_coh(&A.queue[LRG_AQ_RING + 2 * (A.qmask + 1) + (i & A.gmask)],
                           (lrg_gemv_ring_tag(i, A.gmask) << 20) | ((((nt_in & 0xFFF) - 1) & 15) << 16) | ((((nt_nb & 0xFFF) - 1) & 15) << 12) | slot);
            }
            nt_in = __shfl(nt_in, 0); nt_nb = __shfl(nt_nb, 0);
            lrg_async_push_heads(A, slot, nt_in, nt_nb, lane);
        } else if (A.gemv_batch > 1) {
            // batched pooled products: the slot queues up; whoever opens a batch (its first entry) publishes the batch's leader task
            if (lane == 0) {
                const int i = __hip_atomic_fetch_add(&A.queue[LRG_AQ_GTAIL], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lrg_st_coh(&A.queue[LRG_AQ_RING + 2 * (A.qmask + 1) + (i & A.gmask)], (lrg_gemv_ring_tag(i, A.gmask) << 20) | slot);
                if ((i % LRG_GEMV_BATCH) == 0) {
                    const int e = __hip_atomic_fetch_add(&A.queue[LRG_AQ_TAIL + A.head_ring * LRG_AQ_SECOND], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    lrg_st_coh(&A.queue[LRG_AQ_RING + A.head_ring * (A.qmask + 1) + (e & A.qmask)], LRG_TASK(LRG_TASK_GEMV, (i / LRG_GEMV_BATCH) & 0xFFFFF, 0, 64));
                }
            }
        } else {
            const int nb = (A.gemv.C + LRG_GEMV_TASK_COLS - 1) / LRG_GEMV_TASK_COLS;
            lrg_async_push(A, A.head_ring, 2 * nb, lane, [&](int i) { return LRG_TASK(LRG_TASK_GEMV, slot, i / nb, i % nb); });
        }
    }
}

LRG_ASYNC_TASK int lrg_async_task_branch(lrg_kargs_ptr kp_, int code_, int sm_off_, int target_, long long t_task, int &next_ticket, int *ticket_word) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int code = lrg_uniform(code_), sm_off = lrg_uniform(sm_off_), target = lrg_uniform(target_);
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    float *sm = lrg_async_smem + sm_off + LRG_ASYNC_CTL_FLOATS;      // (sm_off: the team's region, control words first)
    const LrgLdsTeam team = lrg_async_team(A, sm, target);
    const int tid = team.tid(), lane = tid & 63;
    const int slot = (code >> 8) & 0xFFFFF, side = (code >> 7) & 1, idx = code & 31, part = (code >> 5) & 3;      // (tile, part of its pooled layer)
    int *word = reinterpret_cast<int *>(sm - LRG_ASYNC_CTL_FLOATS);
    const long r0 = (long)slot * A.front.row_stride + (long)idx * 32;
    long long *stamps = LRG_TRACE ? reinterpret_cast<long long *>(word + 8) : nullptr;      // (LRG_TRACE build: cycle stamps of the tile's phases)
    lrg_fused_tile<32 * 68, 32 * 132, 1, LRG_ASYNC_FD, false, true, true, LrgLdsTeam, true>(A.prob[side], r0, slot, idx, 0x7fffffff, 0x7fffffff, sm, team, stamps,
                                                                                            LrgNoWait(), part, A.branch_parts);
#if LRG_TRACE == 2176
    if (tid == 0 && LRG_DBG(A)) {      // cycles since the tile began, at every stamp (tools/free_run_perf.py prints their means)
        for (int i = 1; i < 21; ++i) if (stamps[i] > stamps[0]) lrg_dbg_add(A, 32 + i, stamps[i] - stamps[0]);
        lrg_dbg_add(A, 32, 1);
    }
#endif
    lrg_exp_delay(LRG_EXP_DELAY_BRANCH);
    // the team's next ticket is on its way while this task's stores drain (an agent-scope atomic with a result is a round trip of its own: taken at the
    // loop's head it was ~1 us between a finished tile and the first look at the next task)
    if (LRG_TICKET_EARLY && ticket_word && tid == 0) next_ticket = __hip_atomic_fetch_add(ticket_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lrg_drain_stores();                              // conv[1] rows and the pooled maxima are out before the arrival
    team.sync();
    if (tid < 64) lrg_async_branch_arrive(A, slot, lane, t_task, true);
    return team.target;
}

// A SHARED tail tile (task index 16; the slot field = the tile's number): rows of several slots, the runs of its 32 rows.  A function of its own (like the pooled
// blocks and the fill-in): inlined beside the two single-slot tiles it cost the worker's loop 150 spill instructions, some inside the tiles' passes.
LRG_ASYNC_ROLE int lrg_async_task_branch_shared(lrg_kargs_ptr kp_, int code_, int sm_off_, int target_, long long t_task, long long t_launch) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int code = lrg_uniform(code_), sm_off = lrg_uniform(sm_off_), target = lrg_uniform(target_);
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    float *sm = lrg_async_smem + sm_off + LRG_ASYNC_CTL_FLOATS;
    const LrgLdsTeam team = lrg_async_team(A, sm, target);
    const int tid = team.tid(), lane = tid & 63;
    const int tile = (code >> 8) & 0xFFFFF, side = (code >> 7) & 1;
    int *word = reinterpret_cast<int *>(sm - LRG_ASYNC_CTL_FLOATS);
    if (tid == 0) word[3] = lrg_tail_wait(A, side, tile, t_launch);      // (the tile was published when its first row was reserved: its rows may still be on their way)
    team.sync();
    const int dead = max(word[3], 0);
    const long r0 = (long)A.front.tail_row0 + (long)tile * 32;
    const int nruns = lrg_fused_tile<32 * 68, 32 * 132, 1, LRG_ASYNC_FD, false, true, true, LrgLdsTeam, false>(A.prob[side], r0, 0, 0, 0x7fffffff, (int)(r0 + 32 - dead), sm, team,
                                                                                                             nullptr, LrgNoWait(), 0, 1);
    lrg_drain_stores();                              // conv[1] rows and the pooled maxima are out before the arrivals
    team.sync();
    // (it arrives for every slot that has rows in it: the runs the tile found, lrg_fused_tile's LDS layout)
    const int *run_inst = reinterpret_cast<const int *>(sm + 32 * 68 + 32 * 132 + 512) + 33;
    if (A.tail_heads && tid == 0) {      // how many slots the tile's HEAD task waits for: known before any of them can be ready (they arrive below)
        int live = 0;
        for (int run = 0; run < nruns; ++run) live += run_inst[run] >= 0 ? 1 : 0;
        const int old = __hip_atomic_fetch_add(lrg_tail_head_word(A, side, tile), live << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" :: "v"(old));      // (performed before the arrivals: the result is waited for)
    }
    team.sync();
    if (tid < 64)
        for (int run = 0; run < nruns; ++run) {
            const int slot = run_inst[run];
            if (slot >= 0) lrg_async_branch_arrive(A, slot, lane, t_task, run == 0);
        }
    team.sync();
    return team.target;
}

// ---- batched pooled products (no units): one trip of a 128-column block of the heads' pooled kernels (512 KB from L2) for up to LRG_GEMV_BATCH slots ----
// With hundreds of slots in flight every slot's pooled product streamed the kernels' 2 MB from L2 by itself: 2.4 TB/s of L2 reads at 1.2 M evaluations a second, 23 us
// per block (7.7 alone) and 17 % of the tile teams' time (profiles/r05_bench_debug_320.log) -- the lock-step iterations do it as ONE GEMM.  Here the slots whose branch
// tiles are all in queue up (the units' ring, unused without units): the first of every LRG_GEMV_BATCH ring positions publishes a LEADER task, whose team waits until
// the batch's entries are written -- or closes it after gemv_batch_ticks (the ring's tail is moved to the batch's end, the unused positions marked) -- and then publishes
// the batch's 2 x 4 block tasks.  A block task computes its 128 columns for every slot of the batch with the arithmetic of lrg_async_gemv, row by row: the same sums in
// the same order, bit for bit.
// The block for all the batch's slots on the matrix cores: lrg_head_gemm_kernel's arithmetic (lrg_net.hip) -- a 32-row x 32-column tile per wavefront (rows = the
// batch's slots, the rest copies of the last one), eight K ranges of P / 8, each a chain of v_mfma_f32_32x32x2_f32 over its k-groups (lane half h feeds k = 8g + 4h + s),
// the eight partial tiles added in order, the bias last: the sums lrg_async_gemv and lrg_head_gemv_kernel compute with FMA chains in that order, bit for bit.  The
// pooled rows come from LDS (staged once, row stride P + 4: conflict-free ds_read_b128), the kernels' columns from L2, half a K range (eight k-groups) ahead of the matrix cores.
template <class TEAM>
__device__ __forceinline__ void lrg_async_gemv_rows(const LrgGemvArgs &g, const int *slots_of, int n, int z, int cb, float *sm, const TEAM &team) {
    const int tid = team.tid(), lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int ldp = g.P + 4;
    float *pl = sm;                                             // [n][P + 4] pooled rows
    for (int j = 0; j < n; ++j)
        for (int i = 2 * tid; i < g.P; i += 2 * FTHREADS) {
            const float2 v = lrg_ld_coh2(g.pooled + (long)slots_of[j] * g.P + i);
            pl[j * ldp + i] = v.x; pl[j * ldp + i + 1] = v.y;
        }
    team.sync();
    const int c0 = cb * LRG_GEMV_TASK_COLS + 32 * wave;
    if (c0 < g.C) {
        const int kq = g.P >> 3;                                // k per range (a multiple of 32: checked by lrg_grow_async)
        const float *arow = pl + min(li, n - 1) * ldp + 4 * lh;
        // The kernels' columns are requested HALF A RANGE ahead: while the 32 MFMAs of one half run, the other half's 32 dwords per lane are on their way (with
        // four k-groups requested at a time and nothing in flight during the MFMAs a block was 32 dependent trips to L2: 200-400 slots lost 40 %).
        f32x16 sum;
        const int hg = kq / 16;                                  // k-groups per half range (8 for P = 1024)
        auto wptr = [&](int half_index) { return g.w[z] + (long)((half_index >> 1) * kq + (half_index & 1) * 8 * hg + 4 * lh) * g.ldw + c0 + li; };
        float w0[8][4], w1[8][4];
        {
            const float *wp = wptr(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float *q = wp + (long)(8 * u) * g.ldw; w0[u][0] = q[0]; w0[u][1] = q[g.ldw]; w0[u][2] = q[2 * (long)g.ldw]; w0[u][3] = q[3 * (long)g.ldw]; }
        }
        f32x16 acc;
        for (int r = 0; r < 8; ++r) {
            const float *ap = arow + r * kq;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            {   // second half of this range on its way, first half on the matrix cores
                const float *wp = wptr(2 * r + 1);
#pragma unroll
                for (int u = 0; u < 8; ++u) { const float *q = wp + (long)(8 * u) * g.ldw; w1[u][0] = q[0]; w1[u][1] = q[g.ldw]; w1[u][2] = q[2 * (long)g.ldw]; w1[u][3] = q[3 * (long)g.ldw]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 av = *reinterpret_cast<const float4 *>(ap + 8 * u);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, w0[u][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, w0[u][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, w0[u][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, w0[u][3], acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (r < 7) {   // first half of the next range on its way, second half of this one on the matrix cores
                const float *wp = wptr(2 * r + 2);
#pragma unroll
                for (int u = 0; u < 8; ++u) { const float *q = wp + (long)(8 * u) * g.ldw; w0[u][0] = q[0]; w0[u][1] = q[g.ldw]; w0[u][2] = q[2 * (long)g.ldw]; w0[u][3] = q[3 * (long)g.ldw]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 av = *reinterpret_cast<const float4 *>(ap + 8 * (hg + u));
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, w1[u][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, w1[u][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, w1[u][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, w1[u][3], acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (r == 0) sum = acc;
            else {
#pragma unroll
                for (int i = 0; i < 16; ++i) sum[i] = sum[i] + acc[i];      // (part[0] + part[1] + ... in order: lrg_head_gemm_kernel's reduction)
            }
        }
        const float bias = g.bias[z] ? g.bias[z][c0 + li] : 0.f;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            if (row < n) lrg_st_coh(g.hb[z] + (long)slots_of[row] * g.C + c0 + li, sum[rr] + bias);
        }
    }
}

// a block of a slot's pooled product is out: the last one publishes the slot's head tiles
template <class TEAM>
__device__ __forceinline__ void lrg_async_gemv_arrive(const LrgAsyncArgs &A, const TEAM &team, int slot, long long t_task) {
    const int tid = team.tid(), lane = tid & 63;
    int32_t *sy = A.sync + (long)slot * LRG_ASYNC_SYNC_WORDS;
    lrg_drain_stores();
    team.sync();
    if (tid < 64) {
        int last = 0, nt_in = 0, nt_nb = 0;
        if (lane == 0) {
            const float4 tq = lrg_ld_coh4(reinterpret_cast<const float *>(sy), 16u);
            const int tgt = __float_as_int(tq.y);
            nt_in = __float_as_int(tq.w) & 0xFFFF; nt_nb = (int)((unsigned)__float_as_int(tq.w) >> 16);
            const int done = __hip_atomic_fetch_add(&sy[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            last = done == tgt;
            if (LRG_DBG(A)) {
                const long long now = wall_clock64();
                lrg_dbg_add(A, 8 + 2 * LRG_TASK_GEMV, now - t_task); lrg_dbg_add(A, 9 + 2 * LRG_TASK_GEMV, 1);
                if (last) lrg_dbg_add(A, 3, (int)((unsigned)now - (unsigned)lrg_ld_coh(&sy[8])));
            }
        }
        if (__shfl(last, 0)) {                       // head 0 = add on the neighbour rows, head 1 = remove on the inlier rows
            nt_in = __shfl(nt_in, 0); nt_nb = __shfl(nt_nb, 0);
            lrg_async_push_heads(A, slot, nt_in, nt_nb, lane);
        }
    }
}

LRG_ASYNC_ROLE int lrg_async_task_gemv(lrg_kargs_ptr kp_, int code_, int sm_off_, int target_, long long t_task) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int code = lrg_uniform(code_), sm_off = lrg_uniform(sm_off_), target = lrg_uniform(target_);
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    float *sm = lrg_async_smem + sm_off + LRG_ASYNC_CTL_FLOATS;      // (sm_off: the team's region, control words first)
    const LrgLdsTeam team = lrg_async_team(A, sm, target);
    const int slot = (code >> 8) & 0xFFFFF, side = (code >> 7) & 1, idx = code & 127;
    lrg_async_gemv(A.gemv, slot, side, idx, sm, team);
    lrg_async_gemv_arrive(A, team, slot, t_task);
    return team.target;
}

// a batch's leader (index 64: waits for / closes the batch, publishes its block tasks) or one of its block tasks (index 32 | block): the slot field = the batch's number
LRG_ASYNC_ROLE int lrg_async_task_gemv_batch(lrg_kargs_ptr kp_, int code_, int sm_off_, int target_, long long t_task, long long t_launch) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int code = lrg_uniform(code_), sm_off = lrg_uniform(sm_off_), target = lrg_uniform(target_);
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    float *sm = lrg_async_smem + sm_off + LRG_ASYNC_CTL_FLOATS;
    const LrgLdsTeam team = lrg_async_team(A, sm, target);
    const int tid = team.tid(), lane = tid & 63;
    const int batch = (code >> 8) & 0xFFFFF, z = (code >> 7) & 1, idx = code & 127;
    const int i0 = batch * LRG_GEMV_BATCH;
    int32_t *ring = A.queue + LRG_AQ_RING + 2 * (A.qmask + 1);
    int *word = reinterpret_cast<int *>(sm - LRG_ASYNC_CTL_FLOATS);      // [8 .. 15]: the batch's slots (LRG_GEMV_NOBODY: an unused position), [3] how many
    if (idx & 64) {
        if (tid == 0) {
            const long long t0 = wall_clock64();
            bool closed = false;
            for (int k = 0; k < LRG_GEMV_BATCH; ++k) {
                const int tag = lrg_gemv_ring_tag(i0 + k, A.gmask);
                for (unsigned spin = 1;; ++spin) {
                    if ((lrg_ld_coh(&ring[(i0 + k) & A.gmask]) >> 20) == tag) break;
                    if (!closed && wall_clock64() - t0 > A.gemv_batch_ticks) {
                        int c = lrg_ld_coh(&A.queue[LRG_AQ_GTAIL]);
                        if (c - i0 >= LRG_GEMV_BATCH) closed = true;      // (every position is taken: their slots are on the way)
                        else if (__hip_atomic_compare_exchange_strong(&A.queue[LRG_AQ_GTAIL], &c, i0 + LRG_GEMV_BATCH, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            for (int q = c; q < i0 + LRG_GEMV_BATCH; ++q) lrg_st_coh(&ring[q & A.gmask], (lrg_gemv_ring_tag(q, A.gmask) << 20) | LRG_GEMV_NOBODY);
                            closed = true;
                        }
                    }
                    if ((spin & 255u) == 0) {
                        if (lrg_ld_coh(&A.queue[LRG_AQ_ABORT])) { k = LRG_GEMV_BATCH; break; }
                        if (wall_clock64() - t_launch > A.abort_ticks) { lrg_st_coh(&A.queue[LRG_AQ_ABORT], 9); k = LRG_GEMV_BATCH; break; }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the marks of a closed batch are out before its block tasks)
        }
        team.sync();
        if (tid < 64) {
            const int nb = (A.gemv.C + LRG_GEMV_TASK_COLS - 1) / LRG_GEMV_TASK_COLS;
            lrg_async_push(A, A.head_ring, 2 * nb, lane, [&](int i) { return LRG_TASK(LRG_TASK_GEMV, batch, i / nb, 32 | (i % nb)); });
        }
        team.sync();
        return team.target;
    }
    // a block task: the batch's entries are final
    if (tid < LRG_GEMV_BATCH) word[8 + tid] = lrg_ld_coh(&ring[(i0 + tid) & A.gmask]) & 0xFFFFF;
    team.sync();
    int members[LRG_GEMV_BATCH], n = 0;
#pragma unroll
    for (int k = 0; k < LRG_GEMV_BATCH; ++k) {
        const int m = word[8 + k];
        if (m != LRG_GEMV_NOBODY) members[n++] = m;
    }
    team.sync();
    if (tid < LRG_GEMV_BATCH) word[8 + tid] = tid < n ? members[tid] : 0;      // (compacted: the rows of the product)
    team.sync();
    lrg_async_gemv_rows(A.gemv, word + 8, n, z, idx & 31, sm, team);
    lrg_drain_stores();
    team.sync();
    if (tid < 64)
        for (int k = 0; k < n; ++k) {
            const int slot = word[8 + k];
            int32_t *sy = A.sync + (long)slot * LRG_ASYNC_SYNC_WORDS;
            int last = 0, nt_in = 0, nt_nb = 0;
            if (lane == 0) {
                const float4 tq = lrg_ld_coh4(reinterpret_cast<const float *>(sy), 16u);
                nt_in = __float_as_int(tq.w) & 0xFFFF; nt_nb = (int)((unsigned)__float_as_int(tq.w) >> 16);
                const int done = __hip_atomic_fetch_add(&sy[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                last = done == __float_as_int(tq.y);
                if (LRG_DBG(A)) {
                    const long long now = wall_clock64();
                    if (k == 0) { lrg_dbg_add(A, 8 + 2 * LRG_TASK_GEMV, now - t_task); lrg_dbg_add(A, 9 + 2 * LRG_TASK_GEMV, 1); }
                    if (last) lrg_dbg_add(A, 3, (int)((unsigned)now - (unsigned)lrg_ld_coh(&sy[8])));
                }
            }
            if (__shfl(last, 0)) {
                nt_in = __shfl(nt_in, 0); nt_nb = __shfl(nt_nb, 0);
                lrg_async_push_heads(A, slot, nt_in, nt_nb, lane);
            }
        }
    team.sync();
    return team.target;
}

// ---- a pooled-product unit (see LRG_GEMV_UNIT_COLS above): `unit` = 0 .. gemv_units - 1, all sixteen wavefronts arrive here ----
LRG_ASYNC_ROLE void lrg_async_gemv_unit(lrg_kargs_ptr kp_, int unit_, long long t_launch) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int unit = lrg_uniform(unit_);
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    const LrgGemvArgs &g = A.gemv;
    const int P = g.P, kq = P >> 3;
    const int upz = g.C / LRG_GEMV_UNIT_COLS;                 // units per head
    const int z = unit / upz, col0 = (unit - z * upz) * LRG_GEMV_UNIT_COLS;
    float *wl = lrg_async_smem;                                // [P][32] this unit's columns of head z's pooled kernel
    {   // the slice, once per launch: 16-byte loads, eight per row
        const float *w = g.w[z] + col0;
        for (int i = threadIdx.x; i < P * (LRG_GEMV_UNIT_COLS / 4); i += LRG_FRONT_THREADS) {
            const int row = i >> 3, q = i & 7;
            *reinterpret_cast<float4 *>(wl + row * LRG_GEMV_UNIT_COLS + 4 * q) = *reinterpret_cast<const float4 *>(w + (long)row * g.ldw + 4 * q);
        }
    }
    const int t = lrg_uniform((int)threadIdx.x >> 8);
    float *sm = lrg_async_smem + P * LRG_GEMV_UNIT_COLS + t * LRG_GEMV_UNIT_TEAM_FLOATS(P);
    float *pl = sm, *part = sm + P;
    int *word = reinterpret_cast<int *>(part + 8 * LRG_GEMV_UNIT_COLS);      // [0], [1] the task of even / odd rounds, [4] barrier counter
    int *ctl = reinterpret_cast<int *>(lrg_async_smem + LRG_GEMV_UNIT_FLOATS(P));          // [0] the unit's next ring entry
    if ((threadIdx.x & 255) == 0) { word[0] = 0; word[1] = 0; word[4] = 0; }
    if (threadIdx.x < 4) ctl[threadIdx.x] = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // the slice is in place; from here on the teams go their own ways
    LrgLdsTeam team;
    team.cnt = &word[4];
    team.target = 0;
    team.base = (int)(threadIdx.x & ~255u);
    team.gave_up = &A.queue[LRG_AQ_ABORT];
    const int tid = team.tid(), lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, r = 2 * wave + (lane >> 5);      // this lane's column and K range
    const float bias = g.bias[z] ? g.bias[z][col0 + (tid & 31)] : 0.f;
    const int32_t *ring = A.queue + LRG_AQ_RING + 2 * (A.qmask + 1);
    for (int round = 0;; round ^= 1) {
        long long t_task = 0;
        if (wave == 0) {
            // the unit's next ring entry, taken by whichever team is free; the other wavefronts wait at the barrier below
            int slot = -2;
            int i = 0;
            if (lane == 0) i = __hip_atomic_fetch_add(&ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            i = __shfl(i, 0);
            const int32_t *e = ring + (i & A.gmask);
            const int tag = lrg_gemv_ring_tag(i, A.gmask);
            for (unsigned spin = 0; slot == -2; ++spin) {
                const int code = lrg_ld_coh(e);
                if ((code >> 20) == tag) { slot = code & 0xFFFFF; break; }      // (slot | tile counts: unpacked below)
                if ((spin & 7) == 7) {
                    if (lrg_ld_coh(&A.queue[LRG_AQ_FRONTS_DONE]) >= A.n_front || lrg_ld_coh(&A.queue[LRG_AQ_ABORT])) slot = -1;
                    else if ((spin & 1023) == 1023 && wall_clock64() - t_launch > A.abort_ticks) {
                        lrg_st_coh(&A.queue[LRG_AQ_ABORT], 4);
                        slot = -1;
                    }
                }
                if (slot == -2) __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0) word[round] = slot;
        }
        team.sync();
        const int entry = word[round];                        // (the other word is written next round: no second barrier)
        if (entry < 0) return;
        const int slot = entry & 0xFFF;                       // (units serve at most LRG_GEMV_UNIT_MAX_SLOTS = 2048 slots)
        if (LRG_DBG(A)) t_task = wall_clock64();
        if (A.pool_rows) {
            // the slot's pooled feature (:122-125) = the maximum over its branch tiles' rows of column maxima, taken here on the way into LDS:
            // every lane owns 16 bytes of the feature and has one load per tile of that side in flight (the values are >= 0: their
            // order is that of their bit patterns, as in the atomicMax formulation)
            const int nt_in = ((entry >> 16) & 15) + 1, nt_nb = ((entry >> 12) & 15) + 1;
            const int half4 = P >> 3;                          // 16-byte pieces per side
            const float *rows = A.pool_rows + (long)slot * A.pool_rows_stride;
            for (int j = tid; j < (P >> 2); j += FTHREADS) {
                const int side = j >= half4 ? 1 : 0, c4 = j - side * half4, nt = side ? nt_nb : nt_in;
                const float *src = rows + (long)side * 16 * (P >> 1);
                int4 m = make_int4(0, 0, 0, 0);
                for (int t0 = 0; t0 < nt; t0 += 8) {          // (eight loads in flight: a side has 3.5 tiles on average, 16 at most)
                    float4 v[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        if (t0 + t < nt) v[t] = lrg_ld_coh4(src, (unsigned)((t0 + t) * (P >> 1) + 4 * c4) * 4u);
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        if (t0 + t < nt) {
                            m.x = max(m.x, __float_as_int(v[t].x)); m.y = max(m.y, __float_as_int(v[t].y));
                            m.z = max(m.z, __float_as_int(v[t].z)); m.w = max(m.w, __float_as_int(v[t].w));
                        }
                }
                *reinterpret_cast<int4 *>(pl + 4 * j) = m;
            }
        } else {   // the slot's pooled row: one 16-byte load per lane
#ifndef LRG_EXP_NO_UNIT_LOAD      // (experiment switch, --policy gt only)
            const float *src = g.pooled + (long)slot * P;
            for (int j = tid; j < (P >> 2); j += FTHREADS)
                *reinterpret_cast<float4 *>(pl + 4 * j) = lrg_ld_coh4(src, (unsigned)j * 16u);
#endif
        }
        team.sync();
        {
            const float *w = wl + (r * kq) * LRG_GEMV_UNIT_COLS + c;
            const float *p = pl + r * kq;
            float acc = 0.f;
            for (int kb = 0; kb < kq; kb += 16) {
                float wv[16];
                float4 pv[4];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = w[(kb + u) * LRG_GEMV_UNIT_COLS];
#pragma unroll
                for (int u = 0; u < 4; ++u) pv[u] = *reinterpret_cast<const float4 *>(p + kb + 4 * u);
                const float pk[16] = {pv[0].x, pv[0].y, pv[0].z, pv[0].w, pv[1].x, pv[1].y, pv[1].z, pv[1].w,
                                      pv[2].x, pv[2].y, pv[2].z, pv[2].w, pv[3].x, pv[3].y, pv[3].z, pv[3].w};
#pragma unroll
                for (int uu = 0; uu < 16; ++uu) {
                    const int u = (uu & 8) | ((uu & 1) << 2) | ((uu >> 1) & 3);      // k = 8g + 0, 4, 1, 5, 2, 6, 3, 7 (lrg_async_gemv)
                    acc = fmaf(pk[u], wv[u], acc);
                }
            }
            part[r * LRG_GEMV_UNIT_COLS + c] = acc;
        }
        team.sync();
        // the sums and the arrival by the LAST wavefront: the first is back at the mailboxes while these stores drain
        if (wave == 3) {
            {   // (32 sums, stored as eight 16-byte pieces)
                float sum = part[lane & (LRG_GEMV_UNIT_COLS - 1)];
#pragma unroll
                for (int q = 1; q < 8; ++q) sum += part[q * LRG_GEMV_UNIT_COLS + (lane & (LRG_GEMV_UNIT_COLS - 1))];
                sum += bias;
                const float v0 = __shfl(sum, 4 * (lane & 7)), v1 = __shfl(sum, 4 * (lane & 7) + 1), v2 = __shfl(sum, 4 * (lane & 7) + 2), v3 = __shfl(sum, 4 * (lane & 7) + 3);
                if (lane < LRG_GEMV_UNIT_COLS / 4) lrg_st_coh4(g.hb[z] + (long)slot * g.C + col0, (unsigned)lane * 16u, make_float4(v0, v1, v2, v3));
            }
            lrg_drain_stores();
            if (lane == 0) {
                int32_t *sy = A.sync + (long)slot * LRG_ASYNC_SYNC_WORDS;
                if (LRG_DBG(A)) {
                    const int done = __hip_atomic_fetch_add(&sy[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                    const long long now = wall_clock64();
                    lrg_dbg_add(A, 8 + 2 * LRG_TASK_GEMV, now - t_task); lrg_dbg_add(A, 9 + 2 * LRG_TASK_GEMV, 1);
                    if (done == lrg_ld_coh(&sy[5])) lrg_dbg_add(A, 3, (int)((unsigned)now - (unsigned)lrg_ld_coh(&sy[8])));
                } else {
                    __hip_atomic_fetch_add(&sy[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// ---- a pooled-product unit of HALF-TEAMS (round 6): eight tasks in flight per unit instead of four ----
// With the register tiles a step is ~69 us and sixteen units x four teams x 3.3 us a task are 81 % busy at 0.99 M evaluations a second: a slot's pooled product
// queues for the slowest of its sixteen units (15 us from the last branch tile to the last sum, profiles/r06_bench_debug_68_reg_tiles.log).  A task's 3.3 us are
// latencies (the ring entry, the slot's pooled row, two barriers, the store, the arrival), not arithmetic, and a fifth team has no place: four staged rows are all the
// LDS holds beside the unit's 128 KB slice.  Here a task is run by TWO wavefronts: the row comes in ONE trip (16 bytes of either half per thread) and is staged half
// at a time (2 KB); a lane owns a column and TWO of the eight K ranges (256 FMAs).  Eight half-teams = sixteen wavefronts: twice the tasks in flight on the same LDS.
// (First form: no staging, the row's values read chunk by chunk with write-through-coherent loads one chunk ahead -- sixteen dependent trips to the memory side per
//  task: 15 us a task, 530 k instance-steps/s.)  The sums are those of lrg_async_gemv_unit bit for bit (per K range a chain
// over k = 8g + 0, 4, 1, 5, 2, 6, 3, 7; the eight partial sums in order; the bias last).
#define LRG_GEMV_UNIT2_TEAM_FLOATS(P) ((P) / 2 + 8 * LRG_GEMV_UNIT_COLS + 32)      // half a pooled row, partial sums, task words + barrier counter
#define LRG_GEMV_UNIT2_FLOATS(P) ((P) * LRG_GEMV_UNIT_COLS + 8 * LRG_GEMV_UNIT2_TEAM_FLOATS(P) + 4)
struct LrgLdsPair {          // two wavefronts meeting at a counter in LDS (LrgLdsTeam's scheme)
    int *cnt;
    mutable int target;
    int *gave_up;
    __device__ __forceinline__ void sync() const {
        target += 2;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0) {
            __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            long long t_long = 0;
            for (unsigned spin = 1; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target; ++spin) {
                if ((spin & 4095u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (!t_long) t_long = now;
                    else if (now - t_long > 500000000LL) { if (gave_up) __hip_atomic_store(gave_up, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
        }
        asm volatile("" ::: "memory");
    }
};

LRG_ASYNC_ROLE void lrg_async_gemv_unit2(lrg_kargs_ptr kp_, int unit_, long long t_launch) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int unit = lrg_uniform(unit_);
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    const LrgGemvArgs &g = A.gemv;
    const int P = g.P, kq = P >> 3;
    const int upz = g.C / LRG_GEMV_UNIT_COLS;
    const int z = unit / upz, col0 = (unit - z * upz) * LRG_GEMV_UNIT_COLS;
    float *wl = lrg_async_smem;                                // [P][32] this unit's columns of head z's pooled kernel
    {
        const float *w = g.w[z] + col0;
        for (int i = threadIdx.x; i < P * (LRG_GEMV_UNIT_COLS / 4); i += LRG_FRONT_THREADS) {
            const int row = i >> 3, q = i & 7;
            *reinterpret_cast<float4 *>(wl + row * LRG_GEMV_UNIT_COLS + 4 * q) = *reinterpret_cast<const float4 *>(w + (long)row * g.ldw + 4 * q);
        }
    }
    const int ht = lrg_uniform((int)threadIdx.x >> 7);
    float *prow_lds = lrg_async_smem + P * LRG_GEMV_UNIT_COLS + ht * LRG_GEMV_UNIT2_TEAM_FLOATS(P);      // [P / 2] half of the slot's pooled row
    float *part = prow_lds + P / 2;                                                                 // [8][32] partial sums
    int *word = reinterpret_cast<int *>(part + 8 * LRG_GEMV_UNIT_COLS);                            // [0], [1] the task of even / odd rounds, [4] barrier counter
    int *ctl = reinterpret_cast<int *>(lrg_async_smem + P * LRG_GEMV_UNIT_COLS + 8 * LRG_GEMV_UNIT2_TEAM_FLOATS(P));      // [0] the unit's next ring entry
    if ((threadIdx.x & 127) == 0) { word[0] = 0; word[1] = 0; word[4] = 0; }
    if (threadIdx.x < 4) ctl[threadIdx.x] = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    LrgLdsPair pair;
    pair.cnt = &word[4]; pair.target = 0; pair.gave_up = &A.queue[LRG_AQ_ABORT];
    const int tid = (int)threadIdx.x & 127, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const float bias = g.bias[z] ? g.bias[z][col0 + c] : 0.f;
    const int32_t *ring = A.queue + LRG_AQ_RING + 2 * (A.qmask + 1);
    for (int round = 0;; round ^= 1) {
        long long t_task = 0;
        if (wave == 0) {
            int slot = -2;
            int i = 0;
            if (lane == 0) i = __hip_atomic_fetch_add(&ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            i = __shfl(i, 0);
            const int32_t *e = ring + (i & A.gmask);
            const int tag = lrg_gemv_ring_tag(i, A.gmask);
            for (unsigned spin = 0; slot == -2; ++spin) {
                const int code = lrg_ld_coh(e);
                if ((code >> 20) == tag) { slot = code & 0xFFFFF; break; }
                if ((spin & 7) == 7) {
                    if (lrg_ld_coh(&A.queue[LRG_AQ_FRONTS_DONE]) >= A.n_front || lrg_ld_coh(&A.queue[LRG_AQ_ABORT])) slot = -1;
                    else if ((spin & 1023) == 1023 && wall_clock64() - t_launch > A.abort_ticks) { lrg_st_coh(&A.queue[LRG_AQ_ABORT], 4); slot = -1; }
                }
                if (slot == -2) __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0) word[round] = slot;
        }
        pair.sync();
        const int entry = word[round];
        if (entry < 0) return;
        const int slot = entry & 0xFFF;
        if (LRG_DBG(A)) t_task = wall_clock64();
        // the slot's pooled row: ONE trip -- every thread of the pair takes 16 bytes of either half (four K ranges each) -- and half a row at a time in LDS: eight
        // staged HALF rows are what fits beside the slice.  Ranges 2 wave + h (first half) and 4 + 2 wave + h (second): each a chain of kq FMAs.
        const float4 ph0 = lrg_ld_coh4(g.pooled + (long)slot * P, (unsigned)tid * 16u), ph1 = lrg_ld_coh4(g.pooled + (long)slot * P, (unsigned)(128 + tid) * 16u);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) pair.sync();                                   // (everybody is done with the first half)
            *reinterpret_cast<float4 *>(prow_lds + 4 * tid) = pass ? ph1 : ph0;
            pair.sync();
            const int rl = 2 * wave + h, r = 4 * pass + rl;
            const float *w = wl + (r * kq) * LRG_GEMV_UNIT_COLS + c;
            const float *p = prow_lds + rl * kq;
            float acc = 0.f;
            for (int kb = 0; kb < kq; kb += 16) {
                float wv[16];
                float4 pv[4];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = w[(kb + u) * LRG_GEMV_UNIT_COLS];
#pragma unroll
                for (int u = 0; u < 4; ++u) pv[u] = *reinterpret_cast<const float4 *>(p + kb + 4 * u);
                const float pk[16] = {pv[0].x, pv[0].y, pv[0].z, pv[0].w, pv[1].x, pv[1].y, pv[1].z, pv[1].w,
                                      pv[2].x, pv[2].y, pv[2].z, pv[2].w, pv[3].x, pv[3].y, pv[3].z, pv[3].w};
#pragma unroll
                for (int uu = 0; uu < 16; ++uu) {
                    const int u = (uu & 8) | ((uu & 1) << 2) | ((uu >> 1) & 3);      // k = 8g + 0, 4, 1, 5, 2, 6, 3, 7 (lrg_async_gemv)
                    acc = fmaf(pk[u], wv[u], acc);
                }
            }
            part[r * LRG_GEMV_UNIT_COLS + c] = acc;
        }
        pair.sync();
        // the sums and the arrival by the second wavefront: the first is back at the ring while these stores drain
        if (wave == 1) {
            float sum = part[lane & (LRG_GEMV_UNIT_COLS - 1)];
#pragma unroll
            for (int q = 1; q < 8; ++q) sum += part[q * LRG_GEMV_UNIT_COLS + (lane & (LRG_GEMV_UNIT_COLS - 1))];
            sum += bias;
            const float v0 = __shfl(sum, 4 * (lane & 7)), v1 = __shfl(sum, 4 * (lane & 7) + 1), v2 = __shfl(sum, 4 * (lane & 7) + 2), v3 = __shfl(sum, 4 * (lane & 7) + 3);
            if (lane < LRG_GEMV_UNIT_COLS / 4) lrg_st_coh4(g.hb[z] + (long)slot * g.C + col0, (unsigned)lane * 16u, make_float4(v0, v1, v2, v3));
            lrg_drain_stores();
            if (lane == 0) {
                int32_t *sy = A.sync + (long)slot * LRG_ASYNC_SYNC_WORDS;
                if (LRG_DBG(A)) {
                    const int done = __hip_atomic_fetch_add(&sy[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                    const long long now = wall_clock64();
                    lrg_dbg_add(A, 8 + 2 * LRG_TASK_GEMV, now - t_task); lrg_dbg_add(A, 9 + 2 * LRG_TASK_GEMV, 1);
                    if (done == lrg_ld_coh(&sy[5])) lrg_dbg_add(A, 3, (int)((unsigned)now - (unsigned)lrg_ld_coh(&sy[8])));
                } else {
                    __hip_atomic_fetch_add(&sy[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// a head tile that was started beside the pooled-product units waits here for its slot's product (lrg_fused_tile, WAIT::late)
struct LrgWaitPooled {
    static constexpr bool late = true;
    const int32_t *sy;           // the slot's arrival counters
    int32_t *queue;
    long long t_launch, abort_ticks;
    __device__ __forceinline__ void operator()() const {
#ifdef LRG_EXP_NO_WAIT_POOLED      // (experiment switch, --policy gt only: what the head tiles' wait for the pooled product costs a step)
        return;
#endif
        const int tgt = lrg_ld_coh(&sy[5]);
        for (unsigned spin = 1; lrg_ld_coh(&sy[1]) < tgt; ++spin) {
            if ((spin & 255u) == 0) {
                if (lrg_ld_coh(&queue[LRG_AQ_ABORT])) break;
                if (wall_clock64() - t_launch > abort_ticks) { lrg_st_coh(&queue[LRG_AQ_ABORT], 5); break; }
            }
            __builtin_amdgcn_s_sleep(LRG_WAIT_POOLED_SLEEP);
        }
    }
};

LRG_ASYNC_TASK int lrg_async_task_head(lrg_kargs_ptr kp_, int code_, int sm_off_, int target_, long long t_task, long long t_launch, int &next_ticket, int *ticket_word) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int code = lrg_uniform(code_), sm_off = lrg_uniform(sm_off_), target = lrg_uniform(target_);
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    float *sm = lrg_async_smem + sm_off + LRG_ASYNC_CTL_FLOATS;      // (sm_off: the team's region, control words first)
    const LrgLdsTeam team = lrg_async_team(A, sm, target);
    const int tid = team.tid();
    const int slot = (code >> 8) & 0xFFFFF, side = (code >> 7) & 1, idx = code & 127;
    int32_t *sy = A.sync + (long)slot * LRG_ASYNC_SYNC_WORDS;
    int *word = reinterpret_cast<int *>(sm - LRG_ASYNC_CTL_FLOATS);
    // (index 16: the head tile of the slot's TAIL rows, where the shared branch tile(s) left their conv[1] rows; the rows behind them are other slots')
    const bool tail = idx == 16;
    const long r0 = tail ? (long)A.front.tail_row0 + lrg_ld_coh(&sy[side ? 9 : 10]) : (long)slot * A.front.row_stride + (long)idx * 32;
    const int rows_out = tail ? ((lrg_ld_coh(&sy[11]) >> (side ? 0 : 16)) & 0xFFFF) : 32;
    long long *stamps = LRG_TRACE ? reinterpret_cast<long long *>(word + 8) : nullptr;
    // (a head tile shares its CU with a branch tile of another slot and is the shorter of the two: issued first where both want a SIMD --
    //  852 -> 856 k instance-steps/s at 68 rooms in flight; the branch tiles first: 847 k)
    __builtin_amdgcn_s_setprio(LRG_ASYNC_HEAD_PRIO);
    if (A.gemv_units) {
        LrgWaitPooled wait;
        wait.sy = sy; wait.queue = A.queue; wait.t_launch = t_launch; wait.abort_ticks = A.abort_ticks;
        lrg_fused_tile<32 * 260, 32 * 68, 1, LRG_ASYNC_FD, false, true, true, LrgLdsTeam, true, LrgWaitPooled>(A.prob[2 + side], r0, slot, tail ? 0 : idx, 0x7fffffff, 0x7fffffff, sm,
                                                                                                               team, stamps, wait, 0, 1, rows_out);
    } else {
        lrg_fused_tile<32 * 260, 32 * 68, 1, LRG_ASYNC_FD, false, true, true, LrgLdsTeam, true>(A.prob[2 + side], r0, slot, tail ? 0 : idx, 0x7fffffff, 0x7fffffff, sm, team, stamps,
                                                                                                LrgNoWait(), 0, 1, rows_out);
    }
#if LRG_TRACE == 8320
    if (tid == 0 && LRG_DBG(A)) {
        for (int i = 1; i < 21; ++i) if (stamps[i] > stamps[0]) lrg_dbg_add(A, 32 + i, stamps[i] - stamps[0]);
        lrg_dbg_add(A, 32, 1);
    }
#endif
    __builtin_amdgcn_s_setprio(0);
    lrg_exp_delay(LRG_EXP_DELAY_HEAD);
    if (LRG_TICKET_EARLY && ticket_word && tid == 0) next_ticket = __hip_atomic_fetch_add(ticket_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lrg_drain_stores();                              // the logits are out before the arrival the front workgroup polls
    team.sync();
    if (tid == 0) {
        const int done = __hip_atomic_fetch_add(&sy[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (LRG_DBG(A)) {
            const long long now = wall_clock64();
            lrg_dbg_add(A, 8 + 2 * LRG_TASK_HEAD, now - t_task); lrg_dbg_add(A, 9 + 2 * LRG_TASK_HEAD, 1);
            if (done == lrg_ld_coh(&sy[6])) { lrg_dbg_add(A, 4, (int)((unsigned)now - (unsigned)lrg_ld_coh(&sy[8]))); lrg_st_coh(&sy[12], (int)(unsigned)now); }
        }
    }
    return team.target;
}

// The head stack on a SHARED tail tile (task index 17; the slot field = the tile's number, `side` = the head): the rows the shared branch tile left their conv[1]
// rows in, every run with its own slot's pooled product as bias (the packed tile form); it arrives for every slot with a run in it.
LRG_ASYNC_ROLE int lrg_async_task_head_shared(lrg_kargs_ptr kp_, int code_, int sm_off_, int target_, long long t_task) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int code = lrg_uniform(code_), sm_off = lrg_uniform(sm_off_), target = lrg_uniform(target_);
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    float *sm = lrg_async_smem + sm_off + LRG_ASYNC_CTL_FLOATS;
    const LrgLdsTeam team = lrg_async_team(A, sm, target);
    const int tid = team.tid();
    const int tile = (code >> 8) & 0xFFFFF, head = (code >> 7) & 1, bside = head ? 0 : 1;      // (head 1 runs on the inlier rows, head 0 on the neighbour rows)
    const int dead = (int)((unsigned)lrg_ld_coh(lrg_tail_word(A, bside, tile)) >> 16);
    const long r0 = (long)A.front.tail_row0 + (long)tile * 32;
    __builtin_amdgcn_s_setprio(LRG_ASYNC_HEAD_PRIO);
    const int nruns = lrg_fused_tile<32 * 260, 32 * 68, 1, LRG_ASYNC_FD, false, true, true, LrgLdsTeam, false>(A.prob[2 + head], r0, 0, 0, 0x7fffffff, (int)(r0 + 32 - dead), sm, team,
                                                                                                             nullptr, LrgNoWait(), 0, 1);
    __builtin_amdgcn_s_setprio(0);
    lrg_drain_stores();                              // the logits are out before the arrivals the front workgroups poll
    team.sync();
    const int *run_inst = reinterpret_cast<const int *>(sm + 32 * 260 + 32 * 68 + 512) + 33;
    if (tid == 0) {
        for (int run = 0; run < nruns; ++run) {
            const int slot = run_inst[run];
            if (slot < 0) continue;
            int32_t *sy = A.sync + (long)slot * LRG_ASYNC_SYNC_WORDS;
            const int done = __hip_atomic_fetch_add(&sy[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            if (LRG_DBG(A)) {
                const long long now = wall_clock64();
                if (run == 0) { lrg_dbg_add(A, 8 + 2 * LRG_TASK_HEAD, now - t_task); lrg_dbg_add(A, 9 + 2 * LRG_TASK_HEAD, 1); }
                if (done == lrg_ld_coh(&sy[6])) lrg_dbg_add(A, 4, (int)((unsigned)now - (unsigned)lrg_ld_coh(&sy[8])));
            }
        }
    }
    team.sync();
    return team.target;
}

// ---- the 1-NN fill-in of a finished room (test_region_grow.py:308-316) as tasks of the same launch ----
// Between launches the fill-ins of the ~13 rooms that finish during a 25 ms launch cost ~2 % of the steady leg (lrg_nn1_fill_batch).  Here the
// front workgroup that finishes a room lists its unlabeled points, makes the room's labels visible (one release: they were plain stores of its CU) and
// publishes one task per 256 candidate points to a ring of their own, served by one team each of a few worker workgroups (fill_wgs: the tile teams'
// tasks are 15-25 us pieces of a slot's critical path and must not queue behind 30 us searches).  A task = lrg_nn1_search_pairs_kernel's work for its
// chunk and ALL unlabeled points: distances in NumPy's pairwise float32 order, first minimum, 64-bit atomicMin per query; the last chunk to arrive writes
// the room's filled labels.  Same labels as the host-launched kernels (tests/test_gpu_free_run.py), and the host no longer has anything to do
// between two launches but to read the statistics block.
template <int FT, class TEAM>
__device__ __forceinline__ void lrg_async_fill_chunk(const float *points, const int32_t *label_in, const int32_t *list, unsigned long long *best, int U,
                                                     int n, int c0, float *sm, const TEAM &team) {
    float *rows = sm;                                              // [128 pairs][FT][2]
    int *lab = reinterpret_cast<int *>(sm + LRG_NN1_C * FT);       // [256] 0 = labeled, -1 = no label / past the end
    unsigned long long *part = reinterpret_cast<unsigned long long *>(sm + ((LRG_NN1_C * FT + LRG_NN1_C + 3) & ~3));      // [4][64]
    const int tid = team.tid(), lane = tid & 63, wave = tid >> 6;
    const int nc = min(LRG_NN1_C, n - c0);
    for (int e = tid; e < LRG_NN1_C * FT; e += FTHREADS) {
        const int r = e / FT, l = e - r * FT;
        rows[(r >> 1) * 2 * FT + 2 * l + (r & 1)] = r < nc ? points[(long)c0 * FT + e] : 0.f;
    }
    for (int r = tid; r < LRG_NN1_C; r += FTHREADS) lab[r] = (r < nc && label_in[c0 + r] != 0) ? 0 : -1;
    constexpr int PER = LRG_NN1_C / 2 / 4;
    for (int q0 = 0; q0 < U; q0 += LRG_NN1_Q) {
        const int qi = list[min(q0 + lane, U - 1)];
        lrg_f2 me2[FT];
#pragma unroll
        for (int l = 0; l < FT; ++l) { const float v = points[(long)qi * FT + l]; me2[l] = lrg_f2{v, v}; }
        team.sync();                                               // rows staged / part[] of the previous round consumed
        unsigned bd = 0xFFFFFFFFu;
        int bi = -1;
        for (int p = wave * PER; p < (wave + 1) * PER; ++p) {
            const int2 lb = *reinterpret_cast<const int2 *>(&lab[2 * p]);
            const lrg_f2 d = lrg_np_sqdist2<FT>(rows + p * 2 * FT, me2);
            const unsigned d0 = __float_as_uint(d.x) | (unsigned)lb.x, d1 = __float_as_uint(d.y) | (unsigned)lb.y;
            bi = d0 < bd ? 2 * p : bi;
            bd = min(bd, d0);
            bi = d1 < bd ? 2 * p + 1 : bi;
            bd = min(bd, d1);
        }
        part[wave * LRG_NN1_Q + lane] = bi >= 0 ? (((unsigned long long)bd << 32) | (unsigned)(c0 + bi)) : ~0ull;
        team.sync();
        if (wave == 0 && q0 + lane < U) {
            const unsigned long long m = min(min(part[lane], part[LRG_NN1_Q + lane]), min(part[2 * LRG_NN1_Q + lane], part[3 * LRG_NN1_Q + lane]));
            if (m != ~0ull) atomicMin(&best[qi], m);
        }
    }
}

LRG_ASYNC_ROLE int lrg_async_task_fill(lrg_kargs_ptr kp_, int code_, int sm_off_, int target_) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int code = lrg_uniform(code_), sm_off = lrg_uniform(sm_off_), target = lrg_uniform(target_);
    const LrgAsyncKArgs &K = LRG_ASYNC_KARGS();
    const LrgAsyncArgs &A = K.A;
    float *sm = lrg_async_smem + sm_off + LRG_ASYNC_CTL_FLOATS;      // (sm_off: the team's region, control words first)
    const LrgLdsTeam team = lrg_async_team(A, sm, target);
    const int tid = team.tid();
    const int room = (code >> 10) & 0x3FFFF, chunk = code & 1023;
    const LrgRoom *R = &K.rooms[room];
    const int n = R->n;
    const long off = R->label - A.fill_label_base;                 // the room's place in the arenas
    const int32_t *label_in = R->label;
    const int32_t *list = A.fill_list + off;
    unsigned long long *best = A.fill_best + off;
    int32_t *fs = A.fill_sync + 4 * (long)room;
    // the room's labels, the list and the reset `best` words were plain stores of the publishing workgroup, released before the task was published
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int U = fs[0];
    if (U > 0) lrg_async_fill_chunk<13>(R->points, label_in, list, best, U, n, chunk * LRG_NN1_C, sm, team);
    lrg_drain_stores();                                            // this chunk's atomicMin are out before the arrival
    team.sync();
    int *word = reinterpret_cast<int *>(sm - LRG_ASYNC_CTL_FLOATS);
    if (tid == 0) {
        const int done = __hip_atomic_fetch_add(&fs[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        word[2] = done == fs[2];
    }
    team.sync();
    if (word[2]) {                                                 // the last chunk: the room's filled labels (:316)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        int32_t *out = A.fill_out_base + off;
        for (int i = tid; i < n; i += FTHREADS) {
            const int li = label_in[i];
            int v = li;
            if (li == 0) {
                const unsigned long long k = __hip_atomic_load(&best[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = k == ~0ull ? 0 : label_in[(int)(k & 0xFFFFFFFFull)];
            }
            out[i] = v;
        }
        if (tid == 0) fs[3] = 1;
    }
    team.sync();
    return team.target;
}

// the front workgroup (all its threads) that has just finished `room`: list of the unlabeled points, release, tasks
__device__ __noinline__ void lrg_async_fill_publish(lrg_kargs_ptr kp_, int room_, long long t_launch) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int room = lrg_uniform(room_);
    const LrgAsyncKArgs &K = LRG_ASYNC_KARGS();
    const LrgAsyncArgs &A = K.A;
    const int tid = threadIdx.x, lane = tid & 63;
    const LrgRoom *R = &K.rooms[room];
    const int n = R->n;
    const long off = R->label - A.fill_label_base;
    const int32_t *label = R->label;
    int32_t *list = A.fill_list + off;
    unsigned long long *best = A.fill_best + off;
    int32_t *fs = A.fill_sync + 4 * (long)room;
    int *cnt = reinterpret_cast<int *>(lrg_async_smem) + 8;        // (LrgFrontShared.flags: free between two front steps)
    if (tid == 0) *cnt = 0;
    __syncthreads();
    for (int i = tid; i < n; i += LRG_FRONT_THREADS) {
        best[i] = ~0ull;
        if (label[i] == 0) list[atomicAdd(cnt, 1)] = i;            // (any order: every query is looked up by its own index)
    }
    __syncthreads();
    const int chunks = (n + LRG_NN1_C - 1) / LRG_NN1_C;
    if (tid == 0) { fs[0] = *cnt; fs[1] = 0; fs[2] = chunks; fs[3] = 0; }
    lrg_drain_stores();                                            // (this wavefront's list / best stores have reached the L2: the barrier alone does not wait for them)
    __syncthreads();                                               // every wavefront's stores are issued ...
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");         // ... and written back (labels, visited marks of this room included)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid < 64) {
        const int fmask = LRG_ASYNC_FILL_RING - 1;
        int32_t *ring = A.queue + LRG_AQ_RING + 2 * (A.qmask + 1) + (A.gmask + 1);
        for (int c0 = 0; c0 < chunks; c0 += 64) {
            const int m = min(64, chunks - c0);
            int base = 0;
            if (lane == 0) {
                // Back-pressure: the ring has a fixed number of entries and rooms may finish faster than their tasks are taken (large scenes, few
                // fill teams).  An entry may be overwritten only when the one a ring's length before it has been taken AND zeroed by its team;
                // tickets are handed out in order and a team zeroes its entry right after reading it, so with at most HALF a ring outstanding
                // (reserved - tickets handed out) that holds with thousands of tasks to spare.  The fill teams depend on nobody: waiting here
                // cannot deadlock, and it is bounded like every other wait of this kernel.
                for (unsigned spin = 1;; ++spin) {
                    const int out = lrg_ld_coh(&A.queue[LRG_AQ_FTAIL]) - lrg_ld_coh(&A.queue[LRG_AQ_FHEAD]);
                    if (out + m <= LRG_ASYNC_FILL_RING / 2 || lrg_ld_coh(&A.queue[LRG_AQ_ABORT])) break;
                    if ((spin & 1023u) == 0 && wall_clock64() - t_launch > A.abort_ticks) { lrg_st_coh(&A.queue[LRG_AQ_ABORT], 7); break; }
                    __builtin_amdgcn_s_sleep(16);
                }
                base = __hip_atomic_fetch_add(&A.queue[LRG_AQ_FTAIL], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            base = __shfl(base, 0);
            if (lane < m) lrg_st_coh(&ring[(base + lane) & fmask], (LRG_TASK_FILL << 28) | (room << 10) | (c0 + lane));
        }
    }
    __syncthreads();
}
// (the caller then parks the slot as idle: a finished room is published once, not again by the next launch that finds the slot still bound to it)

// ---- a worker team: tasks until every front workgroup is done ----
// role_: 0 = by the team's place in its workgroup (below); 1 = a fill-in team (wave-branch mode: wavefronts 4 .. 7 of a wave-branch CU); 2 = ring 1 (wave-branch
// mode: every team of the head teams' CUs -- pooled blocks and head tiles; the branch tiles have rings and wavefronts of their own)
LRG_ASYNC_ROLE void lrg_async_worker(lrg_kargs_ptr kp_, int sm_off_, long long t_launch, int role_) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int sm_off = lrg_uniform(sm_off_), role = lrg_uniform(role_);
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    float *sm = lrg_async_smem + sm_off + LRG_ASYNC_CTL_FLOATS;      // (sm_off: the team's region, control words first)
    LrgLdsTeam team = lrg_async_team(A, sm, 0);
    const int tid = team.tid();
    int *word = reinterpret_cast<int *>(sm - LRG_ASYNC_CTL_FLOATS);       // [0] task of this round
    const int team_no = lrg_uniform((int)threadIdx.x >> 8), wg_no = (int)blockIdx.x - A.worker_base;
    // (small teams have the LDS of a branch tile only: ring 0; beside them every other team runs the rest)
    const bool secondary = role == 2 || (role == 0 && A.head_ring == 1 && (A.small_teams ? team_no >= lrg_async_small_teams(A, wg_no) : 2 * team_no + (wg_no & 1) >= A.ring0_halves));
    const bool filler = role == 1 || (role == 0 && A.fill_list && wg_no < A.fill_wgs && team_no == A.teams - 1 + A.fill_extra);      // this team serves the fill-in ring only (fill_extra: a team MORE on these workgroups)
    // (at 68 slots two branch tiles on a CU slow each other: 809 k -> 783 k instance-steps/s with some second teams on ring 0; at 272 slots
    //  with three teams a single branch team per CU is what every slot queues for: 257 us from publishing to the last branch tile)
    const int ring = secondary ? 1 : 0;      // (more than one team per workgroup: the first teams run the branch tiles, the others the rest)
    // (fill_hybrid: a fill-in team that is a tile team like the others -- a fill-in task when one is waiting [reserved - handed out > 0: the ticket by compare-and-swap, so
    //  that no ticket is taken for a task that does not exist], ring 1's next task otherwise.  A room's ~40-170 fill-in tasks a millisecond left 64 teams of four
    //  hundred idle 98 % of the time while ring 1's tasks queued for theirs.)
    const bool hybrid = filler && role == 0 && A.fill_hybrid && secondary;
    int *ticket_word = &A.queue[filler && !hybrid ? LRG_AQ_FHEAD : LRG_AQ_HEAD + ring * LRG_AQ_SECOND];
    int *early_word = hybrid ? nullptr : ticket_word;
    int next_ticket = -1;                    // (thread 0: the ticket a tile task took while its stores drained)
    bool draining = false;                   // (hybrid, thread 0: every front workgroup has left -- the fill-in tasks that are left, then out)
    for (;;) {
        long long t_task = 0;
        if (tid == 0) {
            const long long t_wait = LRG_DBG(A) ? wall_clock64() : 0;
            bool fill_task = filler && !hybrid;
            int t = 0, code = 0;
            if (hybrid) {      // (its tile tasks take no ticket ahead -- early_word below -- so that none is held while it fills in)
                for (;;) {
                    int head = lrg_ld_coh(&A.queue[LRG_AQ_FHEAD]);
                    const int tail = lrg_ld_coh(&A.queue[LRG_AQ_FTAIL]);
                    if (tail - head <= 0) break;
                    if (__hip_atomic_compare_exchange_strong(&A.queue[LRG_AQ_FHEAD], &head, head + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { fill_task = true; t = head; break; }
                }
                if (!fill_task && draining) code = -1;
            }
            if (code == 0) {
                if (!hybrid || !fill_task) {      // (a team of one ring: its ticket word; hybrid: ring 1's, unless the compare-and-swap above took a fill-in ticket)
                    t = next_ticket >= 0 ? next_ticket : __hip_atomic_fetch_add(ticket_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    next_ticket = -1;
                }
                int *slot = fill_task ? &A.queue[LRG_AQ_RING + 2 * (A.qmask + 1) + (A.gmask + 1) + (t & (LRG_ASYNC_FILL_RING - 1))]
                                      : &A.queue[LRG_AQ_RING + ring * (A.qmask + 1) + (t & A.qmask)];
                for (unsigned spin = 0;; ++spin) {
                    code = lrg_ld_coh(slot);
                    if (code) break;
                    if ((spin & 7) == 7) {
                        if (lrg_ld_coh(&A.queue[LRG_AQ_ABORT])) { code = -1; break; }
                        if (lrg_ld_coh(&A.queue[LRG_AQ_FRONTS_DONE]) >= A.n_front) {
                            // Every front workgroup has left, so nothing more is published.  A tile team's entries are all taken by then (the front
                            // workgroups waited for their results); nobody waits for a fill-in, so a filler leaves only when its ticket lies beyond the
                            // FINAL reservation count -- an entry reserved before the last front workgroup left is written (write-through, drained before
                            // FRONTS_DONE was raised) and is waited for here instead of being given up on a stale look at the slot.
                            // (hybrid: a compare-and-swap ticket is never beyond the count; on a tile ticket it turns to what is left of the fill-in ring)
                            if (hybrid && !fill_task) { code = -2; break; }
                            if (!fill_task || t - lrg_ld_coh(&A.queue[LRG_AQ_FTAIL]) >= 0) { code = -1; break; }
                        }
                        if ((spin & 1023) == 1023 && wall_clock64() - t_launch > A.abort_ticks) {
                            lrg_st_coh(&A.queue[LRG_AQ_ABORT], 2);
                            code = -1;
                            break;
                        }
                    }
                    for (int q = 0; q < A.poll_sleep; ++q) __builtin_amdgcn_s_sleep(LRG_WORKER_POLL_SLEEP);
                }
                if (code > 0) lrg_st_coh(slot, 0);
            }
            if (code == -2) { draining = true; code = 0; }
            word[0] = code;
            t_task = wall_clock64();
            if (LRG_DBG(A)) { lrg_dbg_add(A, 16, t_task - t_wait); lrg_dbg_add(A, 17, 1); }
        }
        team.sync();
        const int code = word[0];                            // (no barrier behind the read: thread 0 writes the next task only after the barriers
        if (code < 0) return;                                //  INSIDE this one, which every wavefront reaches after it has read the word)
        if (code == 0) { team.sync(); continue; }            // (hybrid fill-in team, the launch's end: once more round the fill-in ring)
        const int type = (code >> 28) & 7;
        if (type == LRG_TASK_FILL) team.target = lrg_async_task_fill(kp, code, sm_off, team.target);
        else if (type == LRG_TASK_BRANCH && (code & 31) == 16) team.target = lrg_async_task_branch_shared(kp, code, sm_off, team.target, t_task, t_launch);
        else if (type == LRG_TASK_BRANCH) team.target = lrg_async_task_branch(kp, code, sm_off, team.target, t_task, next_ticket, early_word);
        else if (type == LRG_TASK_GEMV && (code & 96)) team.target = lrg_async_task_gemv_batch(kp, code, sm_off, team.target, t_task, t_launch);
        else if (type == LRG_TASK_GEMV) team.target = lrg_async_task_gemv(kp, code, sm_off, team.target, t_task);
        else if ((code & 127) == 17) team.target = lrg_async_task_head_shared(kp, code, sm_off, team.target, t_task);
        else team.target = lrg_async_task_head(kp, code, sm_off, team.target, t_task, t_launch, next_ticket, early_word);
    }
}

// ---- a wavefront of a wave-branch CU (round 6, lrg_wave_tile.inl): branch tasks (tile, the CU's quarter of the pooled layer) of the CU's side until every front
//      workgroup is done.  The CU's kernels are in LDS (lrg_wave_load_kernels, before the workgroup's only barrier); every wavefront is a worker of its own: its
//      ticket, its ring entry, its task, its arrival -- nothing is shared with the wavefronts beside it but the LDS they read ----
__device__ __forceinline__ void lrg_wave_load_prefix_kernels(const LrgFusedProb &P, int wa, int tid, int nthreads) {      // layers 0 - 3 of one branch
    const int offs[4] = {LRG_WA_W0, LRG_WA_W1, LRG_WA_W2, LRG_WA_W3}, n4[4] = {256, 1024, 1024, 2048};
    const int boff[4] = {LRG_WA_B0, LRG_WA_B1, LRG_WA_B2, LRG_WA_B3}, bn[4] = {LRG_WB_C0, LRG_WB_C1, LRG_WB_C2, LRG_WB_C3};
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const float4 *src = reinterpret_cast<const float4 *>(P.L[l].w);
        float4 *dst = reinterpret_cast<float4 *>(lrg_async_smem + wa + offs[l]);
        for (int i = tid; i < n4[l]; i += nthreads) dst[i] = src[i];
        for (int i = tid; i < bn[l]; i += nthreads) lrg_async_smem[wa + boff[l] + i] = P.L[l].bias[i];
    }
}
__device__ __forceinline__ void lrg_wave_load_pool_kernels(const LrgFusedProb &P, int q, int wp, int tid, int nthreads) {      // quarter q of the pooled layer
    const float4 *src = reinterpret_cast<const float4 *>(P.L[4].w) + (long)q * 4096;      // (the quarter's four column blocks are contiguous in the image)
    float4 *dst = reinterpret_cast<float4 *>(lrg_async_smem + wp + LRG_WP_W4);
    for (int i = tid; i < 4096; i += nthreads) dst[i] = src[i];
    for (int i = tid; i < 128; i += nthreads) lrg_async_smem[wp + LRG_WP_B4 + i] = P.L[4].bias[q * 128 + i];
}

// a wavefront's next task from wave ring `rg` (lane 0 polls; every lane gets the code: < 0 = leave)
__device__ __forceinline__ int lrg_async_wave_take(const LrgAsyncArgs &A, int rg, int &next_ticket, long long t_launch, long long &t_task, int lane) {
    int code = 0;
    if (lane == 0) {
        const long long t_wait = LRG_DBG(A) ? wall_clock64() : 0;
        int *ticket_word = &A.queue[LRG_AQ_WAVE(A) + 32 * rg + 16];
        const int t = next_ticket >= 0 ? next_ticket : __hip_atomic_fetch_add(ticket_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        next_ticket = -1;
        int32_t *e = A.queue + LRG_AQ_WAVE_RING(A, rg) + (t & A.wmask);
        for (unsigned spin = 0;; ++spin) {
            code = lrg_ld_coh(e);
            if (code) break;
            if ((spin & 7) == 7) {
                if (lrg_ld_coh(&A.queue[LRG_AQ_ABORT]) || lrg_ld_coh(&A.queue[LRG_AQ_FRONTS_DONE]) >= A.n_front) { code = -1; break; }
                if ((spin & 1023) == 1023 && wall_clock64() - t_launch > A.abort_ticks) { lrg_st_coh(&A.queue[LRG_AQ_ABORT], 2); code = -1; break; }
            }
            for (int q = 0; q < A.poll_sleep; ++q) __builtin_amdgcn_s_sleep(LRG_WORKER_POLL_SLEEP);
        }
        if (code > 0) lrg_st_coh(e, 0);
        t_task = wall_clock64();
        if (LRG_DBG(A)) { lrg_dbg_add(A, 16, t_task - t_wait); lrg_dbg_add(A, 17, 1); }
    }
    return __builtin_amdgcn_readfirstlane(code);
}

// (both inlined into lrg_grow_async_worker_kernel: a function of its own is compiled for any workgroup size -- 128 VGPRs -- and the tiles need ~150)
// PREFIX tasks: code = LRG_TASK(BRANCH, slot, side, tile)
__device__ __forceinline__ void lrg_async_wave_prefix_worker(lrg_kargs_ptr kp, long long t_launch) {
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    const int lane = (int)threadIdx.x & 63;
    int next_ticket = -1;
    __builtin_amdgcn_s_setprio(2);      // (a fill-in team may share the SIMD: the MFMA-bound wave issues first)
    for (;;) {
        long long t_task = 0;
        const int code = lrg_async_wave_take(A, 4, next_ticket, t_launch, t_task, lane);
        if (code < 0) return;
        const int slot = (code >> 8) & 0xFFFFF, side = (code >> 7) & 1, idx = code & 15;
        const LrgFusedProb &P = A.prob[side];
        const long r0 = (long)slot * A.front.row_stride + (long)idx * 32;
        lrg_wave_prefix_tile(P.x, P.center, P.L[1].gout, A.h3[side], r0, slot, side * LRG_WA_SIDE, lane);
        // the tile's POOL tasks: wave_split / 2 entries in each of the side's two rings, reserved while the rows drain (lanes 0 / 1: the halves)
        const int per_ring = A.wave_split >> 1;
        int base = 0;
        if (lane < 2) base = __hip_atomic_fetch_add(&A.queue[LRG_AQ_WAVE(A) + 32 * (side * 2 + lane)], per_ring, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (LRG_TICKET_EARLY && lane == 2) next_ticket = __hip_atomic_fetch_add(&A.queue[LRG_AQ_WAVE(A) + 32 * 4 + 16], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        next_ticket = __shfl(next_ticket, 2);
        lrg_drain_stores();                              // conv[1] and layer-3 rows are out before anybody is told
        const int half = lane / per_ring, j = lane - half * per_ring;
        const int b = __shfl(base, half & 1);
        if (lane < A.wave_split) {
            const int q = half * 2 + (A.wave_split == 8 ? j >> 1 : j), pair = A.wave_split == 8 ? j & 1 : 0;
            lrg_st_coh(&A.queue[LRG_AQ_WAVE_RING(A, side * 2 + half) + ((b + j) & A.wmask)], LRG_TASK(LRG_TASK_BRANCH, slot, side, idx) | (q << 5) | (pair << 4));
        }
        if (LRG_DBG(A) && lane == 0) { lrg_dbg_add(A, 8 + 2 * LRG_TASK_FILL, wall_clock64() - t_task); lrg_dbg_add(A, 9 + 2 * LRG_TASK_FILL, 1); }      // (index 4's pair: PREFIX tasks)
    }
}

// POOL tasks of ring `rg` = side * 2 + half: code = LRG_TASK(BRANCH, slot, side, tile) | quarter << 5 | pair << 4
__device__ __forceinline__ void lrg_async_wave_pool_worker(lrg_kargs_ptr kp, int rg, long long t_launch) {
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    const int lane = (int)threadIdx.x & 63;
    const int side = rg >> 1;
    const LrgFusedProb &P = A.prob[side];
    int next_ticket = -1;
    __builtin_amdgcn_s_setprio(2);
    for (;;) {
        long long t_task = 0;
        const int code = lrg_async_wave_take(A, rg, next_ticket, t_launch, t_task, lane);
        if (code < 0) return;
        const int slot = (code >> 8) & 0xFFFFF, idx = code & 15, q = (code >> 5) & 3, pair = (code >> 4) & 1;
        const long r0 = (long)slot * A.front.row_stride + (long)idx * 32;
        const int p_lo = A.wave_split == 8 ? pair : 0, p_hi = A.wave_split == 8 ? pair + 1 : 2;
        lrg_wave_pool_tile(A.h3[side], P.pool + (long)slot * P.pool_stride, r0, q, (q & 1) * LRG_WP_QUARTER, p_lo, p_hi, lane);
        if (LRG_TICKET_EARLY && lane == 0) next_ticket = __hip_atomic_fetch_add(&A.queue[LRG_AQ_WAVE(A) + 32 * rg + 16], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        next_ticket = __shfl(next_ticket, 0);
        lrg_drain_stores();                              // the pooled maxima are out before the arrival
        lrg_async_branch_arrive(A, slot, lane, t_task, true);
    }
}

// A slot that leaves the launch between two evaluations: the logits of its tail rows (in the shared rows, lrg_async.inl "shared tail tiles") are copied to the rows of
// its own that the tail would have had without sharing, and its tail bases are cleared -- the next launch hands the shared rows out again from row 0 (its cursors are
// cleared by the host), and the slot's first turn there reads the logits of its last evaluation.  All threads of the slot's front workgroup.
__device__ __forceinline__ void lrg_async_tail_logits_home(const LrgAsyncArgs &A, int s) {
    const LrgFrontArgs &a = A.front;
    const int tid = threadIdx.x;
    if (!a.tail_base) return;
    const int side = tid >> 6, k = tid & 63;                 // (wavefront 0: the inlier side's remove logits, wavefront 1: the neighbour side's add logits)
    if (side < 2) {
        const int tb = a.tail_base[2 * s + side];
        if (tb >= 0) {
            const int rows = a.slot_rows[4 * s + side], first_tail = rows & ~31, tl = rows & 31;
            float *lg = const_cast<float *>(side ? a.add_logits : a.rmv_logits);
            if (k < tl) {
                const float2 v = lrg_ld_coh2(lg + 2 * ((long)a.tail_row0 + tb + k));
                lrg_st_coh2(lg + 2 * ((long)s * a.row_stride + first_tail + k), v.x, v.y);
            }
        }
    }
    lrg_drain_stores();
    __syncthreads();
    if (tid < 2) a.tail_base[2 * s + tid] = -1;
    lrg_drain_stores();
    __syncthreads();
}

struct LrgAsyncFrontCtl {
    int state[LRG_ASYNC_MAX_SERVED];     // 0 to be served, 1 evaluation in flight, 2 finished for this launch; speculation: 3 no seed left for the slot while the
                                         // group's room is not finished (parked until the group is rebound), 4 region pending (served when it is the room's
                                         // earliest in flight, or voided)
    int steps[LRG_ASYNC_MAX_SERVED];
    int tgt[LRG_ASYNC_MAX_SERVED][3];    // running targets of the slot's three arrival counters
    int bc[4];                           // broadcasts of thread 0
    int wb[8];                           // wave-branch mode: the evaluation's first entry in each of the eight wave rings
};

// ---- one slot's front step, a function of its own: the register allocation of lrg_front_greedy_kernel (no spills) instead of the
//      ~230 spill instructions it had inlined into the serving loop below, for ~100 saved / restored registers per step ----
template <bool SPEC>
LRG_ASYNC_ROLE int lrg_async_front_step(lrg_kargs_ptr kp_, int s_) {
    const lrg_kargs_ptr kp = lrg_uniform(kp_);
    const int s = lrg_uniform(s_);
    const LrgAsyncKArgs &K = LRG_ASYNC_KARGS();
    LrgFrontShared &SH = *reinterpret_cast<LrgFrontShared *>(lrg_async_smem);
    const int r = lrg_front_greedy_slot<true, SPEC>(SH, K.slots, K.rooms, K.A.n_slots, K.prm, K.A.front, K.A.big, s);
    lrg_exp_delay(LRG_EXP_DELAY_FRONT);
    // Every wavefront's stores of this turn are performed before anybody goes on: the slot's next turn -- and the serving loop right behind this call -- start with
    // loads of the slot's words by EVERY wavefront (status, list sizes, ...), and a word whose store is still on its way is read as old by some wavefronts and as
    // new by others: they take different branches and meet at different barriers (round 5: seen as a label checksum that differed in 2 of 16 runs).  A barrier
    // alone does not wait for global stores (`s_waitcnt lgkmcnt(0); s_barrier`).
    lrg_drain_stores();
    __syncthreads();
    return r;
}

// ---- a front workgroup: serves the slots f, f + n_front, ... ----
__device__ __forceinline__ void lrg_async_front(lrg_kargs_ptr kp, long long t_launch) {      // (the kernel's own body: no call, no saved registers)
    const LrgAsyncKArgs &K = LRG_ASYNC_KARGS();
    float *smem = lrg_async_smem;
    const LrgAsyncArgs &A = K.A;
    LrgSlot *slots = K.slots;
    LrgRoom *rooms = K.rooms;
    const int tid = threadIdx.x, lane = tid & 63;
    LrgAsyncFrontCtl &C = *reinterpret_cast<LrgAsyncFrontCtl *>(reinterpret_cast<char *>(smem) + ((sizeof(LrgFrontShared) + 15) & ~(size_t)15));
    const int f = blockIdx.x;
    const int spec_k = A.front.spec_k > 1 ? A.front.spec_k : 0;                // speculation: this workgroup serves the group f K .. f K + K - 1 (one room)
    const int n_served = spec_k ? spec_k : (A.n_slots - f + A.n_front - 1) / A.n_front;          // else the slots f, f + n_front, ...
    const int s_first = spec_k ? f * spec_k : f, s_step = spec_k ? 1 : A.n_front;
    const int row_stride = A.front.row_stride;
    const int n_gemv = A.gemv_units ? A.gemv_units : 2 * ((A.gemv.C + LRG_GEMV_TASK_COLS - 1) / LRG_GEMV_TASK_COLS);
    if (tid < LRG_ASYNC_MAX_SERVED) { C.state[tid] = tid < n_served ? 0 : 2; C.steps[tid] = 0; C.tgt[tid][0] = C.tgt[tid][1] = C.tgt[tid][2] = 0; }
    if (tid == 0) reinterpret_cast<LrgFrontShared *>(smem)->upd[0] = 0;      // (the mask update's flag: zero between two updates; the rendezvous' barrier below is ahead of the first)
    // a slot's rows have a fixed place in the row arrays: their tags are written once per launch
    for (int i = 0; i < n_served; ++i) {
        const int s = s_first + i * s_step;
        for (int j = tid; j < row_stride; j += LRG_FRONT_THREADS) {
            lrg_st_coh(&A.front.row_slot_in[(long)s * row_stride + j], s);
            lrg_st_coh(&A.front.row_slot_nb[(long)s * row_stride + j], s);
        }
    }
    lrg_drain_stores();
    // Start rendezvous: front workgroups, units and tile teams wait for each other, so every workgroup of the launch must be running
    // (one per CU).  On a chip the launch has to itself they all start within a microsecond; when something else holds CUs (another
    // process, a kernel of another stream) some never start while these wait -- found out here within the launch budget + 20 ms
    // (LrgAsyncBuffers.start_wait_us: a kernel that leaves within a budget is waited out) and reported (reason 6), not by the hand-overs'
    // bounds seconds later.
    if (tid == 0) {
        const int all = A.total_wgs;      // (both kernels of a wave-branch launch)
        for (unsigned spin = 0; lrg_ld_coh(&A.queue[LRG_AQ_ARRIVED]) < all; ++spin) {
            if (lrg_ld_coh(&A.queue[LRG_AQ_ABORT])) break;
            if ((spin & 15) == 15 && wall_clock64() - t_launch > A.start_ticks) { lrg_st_coh(&A.queue[LRG_AQ_ABORT], 6); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    for (;;) {
        // a hand-over given up anywhere (or this launch far beyond any sane duration): everybody leaves, the host reports it
        // (Measured and dropped, profiles/r05_ab_front_wait.txt: wavefront 0 waiting HERE for the first of the evaluations in flight, lane k on slot k's counter,
        //  instead of passes of this loop -- 17 looks in vain per evaluation became 0.8 and nothing else moved: a result is seen 4.4 us after its last head tile on
        //  average because the workgroup is inside its OTHER slot's step a fifth of the time (two slots per front workgroup at 68 slots; histogram in
        //  profiles/r05_bench_debug_68.log), and 68 front workgroups cost more worker CUs than that is worth: profiles/r04_fronts_teams_sweep.txt.)
        if (tid == 0) {
            int ab = lrg_ld_coh(&A.queue[LRG_AQ_ABORT]);
            if (!ab && wall_clock64() - t_launch > A.abort_ticks) { ab = 1; lrg_st_coh(&A.queue[LRG_AQ_ABORT], 1); }
            C.bc[2] = ab;
        }
        __syncthreads();
        const int aborted = C.bc[2];
        __syncthreads();
        if (aborted) {
            if (tid == 0 && A.front.stats)      // (count of front workgroups that left this way; the reasons seen, one byte each, above it)
                atomicAdd(reinterpret_cast<unsigned long long *>(&A.front.stats[3]), 1ULL | ((unsigned long long)(aborted & 0xFF) << 32));
            break;
        }
        int live = 0;
        int chosen = -1;
        if (spec_k) {
            // Speculation: ONE slot per pass, the ready one with the earliest seed position -- the room's earliest region in flight is what every later one waits
            // for (in-order commit), so its steps come first; the slots that only speculate take the front workgroup's time that is left.
            if (tid == 0) {
                int first = -1, first_pos = INT_MAX;
                for (int k = 0; k < n_served; ++k) {
                    const int p = slots[s_first + k].spec_pos;
                    if (p < first_pos) { first = k; first_pos = p; }
                }
                const bool over = wall_clock64() - t_launch > A.budget_ticks;
                int best = -1, best_pos = 0, nlive = 0;
                for (int k = 0; k < n_served; ++k) {
                    const int stk = C.state[k];
                    if (stk == 2 || stk == 3) continue;
                    ++nlive;
                    bool ready = stk == 0;
                    if (stk == 1) ready = lrg_ld_coh(&A.sync[(long)(s_first + k) * LRG_ASYNC_SYNC_WORDS + 2]) >= C.tgt[k][2];
                    if (stk == 4) ready = k == first || (slots[s_first + k].spec_flags & 1) || over || (first >= 0 && C.state[first] == 2);
                    const int p = slots[s_first + k].spec_pos;
                    if (ready && (best < 0 || p < best_pos)) { best = k; best_pos = p; }
                }
                C.bc[0] = best; C.bc[1] = nlive;
            }
            __syncthreads();
            chosen = C.bc[0];
            live = C.bc[1];
            __syncthreads();
            if (chosen < 0) {
                if (!live) break;
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
        }
        for (int ii = 0; ii < (spec_k ? 1 : n_served); ++ii) {
            const int i = spec_k ? chosen : ii;
            const int s = s_first + i * s_step;
            int st = C.state[i];
            if (st == 2 || st == 3) continue;
            if (st == 4) {
                // a pending region: its turn has come when no slot of the group holds an earlier seed position, or it was voided; given up for this
                // launch with the budget, or when nobody is left who could commit before it (every other slot finished for this launch)
                if (tid == 0) {
                    // (the slot that holds the room's earliest seed position: it is what this one waits for -- if THAT slot is finished for this launch,
                    //  so is this one; a test for "anybody still at work" gave up while the earliest region's slot was itself pending, a moment before its
                    //  turn, and blocked everything behind it until the next launch)
                    const int my = slots[s].spec_pos;
                    int first = i, first_pos = my;
                    for (int k = 0; k < n_served; ++k) {
                        const int p = slots[s_first + k].spec_pos;
                        if (k != i && p < first_pos) { first = k; first_pos = p; }
                    }
                    C.bc[0] = (first == i || (slots[s].spec_flags & 1)) ? 1 : (C.state[first] == 2 || wall_clock64() - t_launch > A.budget_ticks) ? 2 : 0;
                }
                __syncthreads();
                const int turn = C.bc[0];
                __syncthreads();
                if (turn == 2) { if (tid == 0) C.state[i] = 2; continue; }
                ++live;
                if (turn == 0) continue;
                if (tid == 0) C.state[i] = 0;
                st = 0;
            } else {
                ++live;
            }
            if (st == 1) {
                if (tid == 0) C.bc[0] = lrg_ld_coh(&A.sync[(long)s * LRG_ASYNC_SYNC_WORDS + 2]) >= C.tgt[i][2];
                __syncthreads();
                const int ready = C.bc[0];
                __syncthreads();
                if (!ready) {
                    if (LRG_DBG(A) && tid == 0) lrg_dbg_add(A, 28, 1);      // (looks at a result that was not there yet)
                    continue;
                }
                st = 0;
                if (LRG_DBG(A) && tid == 0) {
                    const unsigned now = (unsigned)wall_clock64();
                    lrg_dbg_add(A, 5, (int)(now - (unsigned)lrg_ld_coh(&A.sync[(long)s * LRG_ASYNC_SYNC_WORDS + 8])));
                    lrg_dbg_add(A, 6, 1);
                    // (how long ago the last head tile arrived, by its own stamp: a histogram in powers of two of 0.5 us)
                    const int ago = (int)(now - (unsigned)lrg_ld_coh(&A.sync[(long)s * LRG_ASYNC_SYNC_WORDS + 12]));
                    int b = 0;
                    for (int lim = 50; b < 7 && ago >= lim; lim *= 2) ++b;
                    lrg_dbg_add(A, 54 + b, 1);
                    lrg_dbg_add(A, 62, ago > 0 && ago < 100000 ? ago : 0);
                }
            }
            // a new evaluation only within the budget of this launch
            if (tid == 0) {
                const long long el = wall_clock64() - t_launch;
                C.bc[1] = C.steps[i] >= A.max_steps || el > A.budget_ticks;
            }
            __syncthreads();
            const int stop = C.bc[1];
            __syncthreads();
            if (stop) {
                // (shared tail tiles: the slot leaves the launch with the logits of its tail rows in rows that the NEXT launch hands out again from row 0 -- its
                //  first turn there would read them while other slots' head tiles may already write into them.  They go home, to the slot's own rows, before it
                //  leaves: its next update finds them there, tail_base = -1.)
#ifndef LRG_EXP_NO_TAIL_HOME      // (experiment switch: the hazard as it was up to round 5 -- tests/test_gpu_free_run.py::test_shared_tail_tiles_across_short_launches fails with it)
                if (A.tail) lrg_async_tail_logits_home(A, s);
#endif
                if (tid == 0) C.state[i] = 2;
                continue;
            }
            const long long t_front = LRG_DBG(A) ? wall_clock64() : 0;
            const int r = __builtin_amdgcn_readfirstlane(spec_k ? lrg_async_front_step<true>(kp, s) : lrg_async_front_step<false>(kp, s));
            if (r == 0) {
                // no evaluation: the slot is idle / its room finished (-> finished for this launch), or it stopped a region / goes on
                // looking for a seed (-> served again at once)
                __syncthreads();
                if (A.fill_list && slots[s].room >= 0 && slots[s].status == LRG_DONE) {      // (uniform: plain loads of the slot this workgroup owns)
                    lrg_async_fill_publish(kp, slots[s].room, t_launch);       // the room is finished: its fill-in as tasks of this launch
                    if (tid == 0) slots[s].status = LRG_IDLE;
                    __syncthreads();
                }
                if (tid == 0) {
                    const int status = slots[s].status;
                    // (speculation: a slot without a seed left is parked while the group's room is unfinished -- IDLE with the room bound; the slot that
                    //  finishes the room went through DONE above and is IDLE with the others now: the group takes its next room as a whole)
                    const bool parked = spec_k && status == LRG_IDLE && slots[s].room >= 0 && !rooms[slots[s].room].done;
                    const bool pending = spec_k && status == LRG_PENDING;
                    const bool idle = !parked && (slots[s].room < 0 || status == LRG_DONE || status == LRG_IDLE);
                    int next = -1;
                    if (idle && A.room_queue) {      // the slot's room is finished (or it has none): the next room of the queue, if any is left
                        if (lrg_ld_coh(&A.room_queue[0]) < A.room_queue[1]) {
                            const int k = __hip_atomic_fetch_add(&A.room_queue[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (k < A.room_queue[1]) next = A.room_queue[2 + k];
                        }
                    }
                    C.bc[3] = next;
                    C.state[i] = parked ? 3 : pending ? 4 : (idle && next < 0) ? 2 : 0;
                    if (spec_k && idle)              // the whole group: bound to the next room together, or finished for this launch together
                        for (int k = 0; k < n_served; ++k) C.state[k] = next < 0 ? 2 : 0;
                }
                __syncthreads();
                const int next = C.bc[3];
                if (next >= 0) lrg_bind_group_device(slots, rooms, spec_k ? s_first : s, spec_k ? spec_k : 1, next & 0x3FFFFFFF, next >> 30, 1);
                __syncthreads();
                continue;
            }
            // The evaluation's targets and the reservation of its queue entries go out with the rows (one trip, not three in a row);
            // rows, centre, the zeroed pooled feature and the targets are out (write-through) once every wavefront has drained; only
            // then the entries are written -- a consumer waits for its entry, not for the reservation.
            // (shared tail tiles: a side whose tail went to the shared rows has tiles of its own for its FULL tiles only; its tail is in one or two shared branch
            //  tiles, which arrive on the slot's counter like its own, and in one head tile of the slot's own on those rows)
            const LrgFrontShared &SHr = *reinterpret_cast<const LrgFrontShared *>(smem);
            const int rin = r >> 16, rnb = r & 0xFFFF;
            const int tb_in = A.tail ? SHr.tail[0] : -1, tb_nb = A.tail ? SHr.tail[1] : -1;
            const int nt_in = tb_in >= 0 ? rin >> 5 : (rin + 31) >> 5, nt_nb = tb_nb >= 0 ? rnb >> 5 : (rnb + 31) >> 5;
            const int sh_in = tb_in >= 0 ? 1 : 0, sh_nb = tb_nb >= 0 ? 1 : 0;
            const int nsh = (sh_in ? 1 + ((((tb_in + (rin & 31) - 1) >> 5) != (tb_in >> 5)) ? 1 : 0) : 0) + (sh_nb ? 1 + ((((tb_nb + (rnb & 31) - 1) >> 5) != (tb_nb >> 5)) ? 1 : 0) : 0);
            // the shared tile whose FIRST row this reservation took (at most one per side): its task goes out with the slot's own
            const int op_in = !sh_in ? -1 : (tb_in & 31) == 0 ? tb_in >> 5 : (((tb_in + (rin & 31) - 1) >> 5) != (tb_in >> 5)) ? (tb_in >> 5) + 1 : -1;
            const int op_nb = !sh_nb ? -1 : (tb_nb & 31) == 0 ? tb_nb >> 5 : (((tb_nb + (rnb & 31) - 1) >> 5) != (tb_nb >> 5)) ? (tb_nb >> 5) + 1 : -1;
            const int n_open = (op_in >= 0 ? 1 : 0) + (op_nb >= 0 ? 1 : 0);
            if (tid == 0) {
                int32_t *sy = A.sync + (long)s * LRG_ASYNC_SYNC_WORDS;
                C.tgt[i][0] += (nt_in + nt_nb) * A.branch_parts + nsh; C.tgt[i][1] += n_gemv; C.tgt[i][2] += nt_in + nt_nb + (A.tail_heads ? nsh : sh_in + sh_nb);
                if (A.tail) { lrg_st_coh(&sy[9], tb_in); lrg_st_coh(&sy[10], tb_nb); lrg_st_coh(&sy[11], (rin & 31) | ((rnb & 31) << 16)); }
                lrg_st_coh4(reinterpret_cast<float *>(sy), 16u, make_float4(__int_as_float(C.tgt[i][0]), __int_as_float(C.tgt[i][1]), __int_as_float(C.tgt[i][2]),
                                                                      __int_as_float((nt_in | (sh_in << 12)) | ((nt_nb | (sh_nb << 12)) << 16))));      // (one 16-byte store instead of five dwords)
                if (LRG_DBG(A)) {
                    const long long now = wall_clock64();
                    lrg_st_coh(&sy[8], (int)(unsigned)now);
                    lrg_dbg_add(A, 0, now - t_front); lrg_dbg_add(A, 1, 1);
                }
                if (A.work) {
                    atomicAdd(&A.work[0], 1ULL); atomicAdd(&A.work[1], (unsigned long long)(r >> 16));
                    atomicAdd(&A.work[2], (unsigned long long)(r & 0xFFFF)); atomicAdd(&A.work[3], (unsigned long long)(nt_in + nt_nb + n_open));
                    atomicAdd(&A.work[7], (unsigned long long)(nt_in + nt_nb + (A.tail_heads ? 0 : sh_in + sh_nb)));      // (head tiles; shared ones are counted where they are published)
                }
                C.state[i] = 1;
                C.steps[i] += 1;
                if (!A.wave_wgs) C.bc[3] = __hip_atomic_fetch_add(&A.queue[LRG_AQ_TAIL], (nt_in + nt_nb) * A.branch_parts + n_open, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (wave-branch mode: the tiles' PREFIX tasks, one reservation in ring 4 -- beside thread 0's stores, by the second wavefront)
            if (A.wave_wgs && tid == 64) C.wb[0] = __hip_atomic_fetch_add(&A.queue[LRG_AQ_WAVE(A) + 32 * 4], nt_in + nt_nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lrg_drain_stores();
            __syncthreads();
            if (A.wave_wgs) {
                if (tid < nt_in + nt_nb)                         // (up to 32 tiles)
                    lrg_st_coh(&A.queue[LRG_AQ_WAVE_RING(A, 4) + ((C.wb[0] + tid) & A.wmask)],
                               tid < nt_in ? LRG_TASK(LRG_TASK_BRANCH, s, 0, tid) : LRG_TASK(LRG_TASK_BRANCH, s, 1, tid - nt_in));
            } else if (tid < 128) {                              // (up to 32 tiles x 4 parts)
                const int nt = nt_in + nt_nb, t = tid / A.branch_parts, part = tid - t * A.branch_parts;
                if (tid < nt * A.branch_parts)
                    lrg_st_coh(&A.queue[LRG_AQ_RING + ((C.bc[3] + tid) & A.qmask)],      // (ring 0)
                               (t < nt_in ? LRG_TASK(LRG_TASK_BRANCH, s, 0, t) : LRG_TASK(LRG_TASK_BRANCH, s, 1, t - nt_in)) | (part << 5));
            }
            if (A.tail && tid >= 128 && tid < 132) {
                // the tail rows are out (drained above): counted into their shared tiles by atomics nobody waits for (lanes 0 / 1: the inlier side's first / second
                // tile, 2 / 3: the neighbour side's); the tasks of the tiles this slot opened, behind its own in the ring
                const int side = (tid - 128) >> 1, second = (tid - 128) & 1;
                const int tb = side ? tb_nb : tb_in, tl = (side ? rnb : rin) & 31;
                if (tb >= 0) {
                    const int ta = tb >> 5, tz = (tb + tl - 1) >> 5, ca = min(32 - (tb & 31), tl);
                    if (!second) lrg_tail_account(A, side, ta, ca, 0);
                    else if (tz != ta) lrg_tail_account(A, side, tz, tl - ca, 0);
                } else if (!second) {
                    const int dead = SHr.tail[2 + side];
                    if (dead) lrg_tail_account(A, side, A.tail_tiles - 1, dead, dead);      // (a reservation that fell off the end of the shared rows)
                }
                const int op = side ? op_nb : op_in;
                if (!second && op >= 0)
                    lrg_st_coh(&A.queue[LRG_AQ_RING + ((C.bc[3] + (nt_in + nt_nb) * A.branch_parts + (side && op_in >= 0 ? 1 : 0)) & A.qmask)], LRG_TASK(LRG_TASK_BRANCH, op, side, 16));
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

__global__ __launch_bounds__(LRG_FRONT_THREADS) void lrg_grow_async_kernel(LrgAsyncKArgs K) {
    const int tid = threadIdx.x;
    const long long t_launch = wall_clock64();
#if defined(__HIP_DEVICE_COMPILE__)
    lrg_kargs_ptr kp = (lrg_kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
#else
    lrg_kargs_ptr kp = nullptr;
#endif
    if (tid == 0) __hip_atomic_fetch_add(&K.A.queue[LRG_AQ_ARRIVED], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (nobody waits for the result)
    if ((int)blockIdx.x >= K.A.n_front && (int)blockIdx.x < K.A.n_front + K.A.gemv_units) {
        // (half-teams: eight tasks in flight per unit; not with per-tile pool rows, whose maximum is taken while the row is staged)
        if (K.A.unit_pairs && !K.A.pool_rows) lrg_async_gemv_unit2(kp, (int)blockIdx.x - K.A.n_front, t_launch);
        else lrg_async_gemv_unit(kp, (int)blockIdx.x - K.A.n_front, t_launch);
        return;
    }
    if ((int)blockIdx.x >= K.A.n_front) {
        // worker workgroup: teams of four consecutive wavefronts (one per SIMD), each on its own part of the LDS
        // (a wave-branch launch has none of these in this kernel: its grid ends with the units)
        const int t = tid >> 8;
        if (t >= K.A.teams + (((int)blockIdx.x - K.A.worker_base < K.A.fill_wgs && K.A.fill_list) ? K.A.fill_extra : 0)) return;
        const int small = lrg_async_small_teams(K.A, (int)blockIdx.x - K.A.worker_base);
        const int sm_off = t < small ? t * LRG_ASYNC_SMALL_TEAM_FLOATS : small * LRG_ASYNC_SMALL_TEAM_FLOATS + (t - small) * LRG_ASYNC_TEAM_FLOATS;
        int *word = reinterpret_cast<int *>(lrg_async_smem + sm_off);
        if ((tid & 255) == 0) { word[4] = 0; word[5] = 0; }  // the team's barrier counter, its 'at work' flag
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // the only workgroup-wide barrier of a worker: before any team has started
        lrg_async_worker(kp, sm_off, t_launch, 0);
        return;
    }
    lrg_async_front(kp, t_launch);
}

// ---- team 0 of a register-tile CU (LrgAsyncArgs.reg_tiles): branch tiles of ring 0, one per turn of the team (inlined into the worker kernel: ~200 VGPRs) ----
#define LRG_RT_TEAM0_FLOATS (LRG_ASYNC_CTL_FLOATS + LRG_RT_XCH_FLOATS)
__device__ __forceinline__ void lrg_async_reg_tile_team(lrg_kargs_ptr kp, long long t_launch, int region) {      // region: LDS offset (floats) of the team's control words
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    float *sm = lrg_async_smem + region + LRG_ASYNC_CTL_FLOATS;
    LrgLdsTeam team = lrg_async_team(A, sm, 0);
    const int tid = team.tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int *word = reinterpret_cast<int *>(sm - LRG_ASYNC_CTL_FLOATS);
    int *ticket_word = &A.queue[LRG_AQ_HEAD];
    int next_ticket = -1;
    for (;;) {
        long long t_task = 0;
        if (tid == 0) {
            const long long t_wait = LRG_DBG(A) ? wall_clock64() : 0;
            const int t = next_ticket >= 0 ? next_ticket : __hip_atomic_fetch_add(ticket_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            next_ticket = -1;
            int *e = &A.queue[LRG_AQ_RING + (t & A.qmask)];
            int code = 0;
            for (unsigned spin = 0;; ++spin) {
                code = lrg_ld_coh(e);
                if (code) break;
                if ((spin & 7) == 7) {
                    if (lrg_ld_coh(&A.queue[LRG_AQ_ABORT]) || lrg_ld_coh(&A.queue[LRG_AQ_FRONTS_DONE]) >= A.n_front) { code = -1; break; }
                    if ((spin & 1023) == 1023 && wall_clock64() - t_launch > A.abort_ticks) { lrg_st_coh(&A.queue[LRG_AQ_ABORT], 2); code = -1; break; }
                }
                for (int q = 0; q < A.poll_sleep; ++q) __builtin_amdgcn_s_sleep(LRG_WORKER_POLL_SLEEP);
            }
            if (code > 0) lrg_st_coh(e, 0);
            word[0] = code;
            t_task = wall_clock64();
            if (LRG_DBG(A)) { lrg_dbg_add(A, 16, t_task - t_wait); lrg_dbg_add(A, 17, 1); }
        }
        team.sync();
        const int code = word[0];                            // (thread 0 writes the next one behind the barriers of the tile)
        if (code < 0) return;
        const int slot = (code >> 8) & 0xFFFFF, side = (code >> 7) & 1, idx = code & 31, part = (code >> 5) & 3;
        const LrgFusedProb &P = A.prob[side];
        const long r0 = (long)slot * A.front.row_stride + (long)idx * 32;
        if (A.branch_parts == 2)
            lrg_team_branch_tile_reg<2>(P.x, P.center, P.L[1].gout, P.pool + (long)slot * P.pool_stride, P.L[3].w, P.L[4].w, r0, slot, side * LRG_RT_SIDE,
                                        region + LRG_ASYNC_CTL_FLOATS, team, wave, lane, part);
        else
            lrg_team_branch_tile_reg<1>(P.x, P.center, P.L[1].gout, P.pool + (long)slot * P.pool_stride, P.L[3].w, P.L[4].w, r0, slot, side * LRG_RT_SIDE,
                                        region + LRG_ASYNC_CTL_FLOATS, team, wave, lane, 0);
        if (LRG_TICKET_EARLY && tid == 0) next_ticket = __hip_atomic_fetch_add(ticket_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lrg_drain_stores();                                  // conv[1] rows and the pooled maxima are out before the arrival
        team.sync();
        if (tid < 64) lrg_async_branch_arrive(A, slot, lane, t_task, true);
    }
}
// ---- team 1 of a register-tile CU: ring 1 -- head tiles as REGISTER HEAD TILES (lrg_team_head_tile_reg), pooled blocks where there are no units (inlined into
//      the worker kernel: ~240 VGPRs) ----
#define LRG_RT_TEAM1_FLOATS (LRG_ASYNC_CTL_FLOATS + LRG_RH_FLOATS)
__device__ __forceinline__ void lrg_async_reg_head_team(lrg_kargs_ptr kp, long long t_launch) {
    const LrgAsyncArgs &A = LRG_ASYNC_KARGS().A;
    const int sm_off = LRG_RT_WEIGHT_FLOATS + LRG_RT_TEAM0_FLOATS;
    float *sm = lrg_async_smem + sm_off + LRG_ASYNC_CTL_FLOATS;
    LrgLdsTeam team = lrg_async_team(A, sm, 0);
    const int tid = team.tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int *word = reinterpret_cast<int *>(sm - LRG_ASYNC_CTL_FLOATS);
    int *ticket_word = &A.queue[LRG_AQ_HEAD + LRG_AQ_SECOND];
    int next_ticket = -1;
    for (;;) {
        long long t_task = 0;
        if (tid == 0) {
            const long long t_wait = LRG_DBG(A) ? wall_clock64() : 0;
            const int t = next_ticket >= 0 ? next_ticket : __hip_atomic_fetch_add(ticket_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            next_ticket = -1;
            int *e = &A.queue[LRG_AQ_RING + (A.qmask + 1) + (t & A.qmask)];
            int code = 0;
            for (unsigned spin = 0;; ++spin) {
                code = lrg_ld_coh(e);
                if (code) break;
                if ((spin & 7) == 7) {
                    if (lrg_ld_coh(&A.queue[LRG_AQ_ABORT]) || lrg_ld_coh(&A.queue[LRG_AQ_FRONTS_DONE]) >= A.n_front) { code = -1; break; }
                    if ((spin & 1023) == 1023 && wall_clock64() - t_launch > A.abort_ticks) { lrg_st_coh(&A.queue[LRG_AQ_ABORT], 2); code = -1; break; }
                }
                for (int q = 0; q < A.poll_sleep; ++q) __builtin_amdgcn_s_sleep(LRG_WORKER_POLL_SLEEP);
            }
            if (code > 0) lrg_st_coh(e, 0);
            word[0] = code;
            t_task = wall_clock64();
            if (LRG_DBG(A)) { lrg_dbg_add(A, 16, t_task - t_wait); lrg_dbg_add(A, 17, 1); }
        }
        team.sync();
        const int code = word[0];
        if (code < 0) return;
        if (((code >> 28) & 7) == LRG_TASK_GEMV) {            // (no units: a 128-column block of a slot's pooled product -- the team tiles' task as it is)
            team.target = lrg_async_task_gemv(kp, code, sm_off, team.target, t_task);
            continue;
        }
        const int slot = (code >> 8) & 0xFFFFF, side = (code >> 7) & 1, idx = code & 127;
        int32_t *sy = A.sync + (long)slot * LRG_ASYNC_SYNC_WORDS;
        const long r0 = (long)slot * A.front.row_stride + (long)idx * 32;
        __builtin_amdgcn_s_setprio(LRG_ASYNC_HEAD_PRIO);
        if (A.gemv_units) {
            LrgWaitPooled wait;
            wait.sy = sy; wait.queue = A.queue; wait.t_launch = t_launch; wait.abort_ticks = A.abort_ticks;
            lrg_team_head_tile_reg(A.prob[2 + side], r0, slot, sm_off + LRG_ASYNC_CTL_FLOATS, team, wait, wave, lane);
        } else {
            lrg_team_head_tile_reg(A.prob[2 + side], r0, slot, sm_off + LRG_ASYNC_CTL_FLOATS, team, LrgNoWait(), wave, lane);
        }
        __builtin_amdgcn_s_setprio(0);
        if (LRG_TICKET_EARLY && tid == 0) next_ticket = __hip_atomic_fetch_add(ticket_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lrg_drain_stores();                                  // the logits are out before the arrival the front workgroup polls
        team.sync();
        if (tid == 0) {
            const int done = __hip_atomic_fetch_add(&sy[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            if (LRG_DBG(A)) {
                const long long now = wall_clock64();
                lrg_dbg_add(A, 8 + 2 * LRG_TASK_HEAD, now - t_task); lrg_dbg_add(A, 9 + 2 * LRG_TASK_HEAD, 1);
                if (done == lrg_ld_coh(&sy[6])) { lrg_dbg_add(A, 4, (int)((unsigned)now - (unsigned)lrg_ld_coh(&sy[8]))); lrg_st_coh(&sy[12], (int)(unsigned)now); }
            }
        }
    }
}
__device__ __forceinline__ void lrg_rt_load_kernels(const LrgFusedProb &P, int ws, int tid, int nthreads) {      // layers 0 - 2 and every layer's bias of one branch
    const int offs[3] = {LRG_RT_W0, LRG_RT_W1, LRG_RT_W2}, n4[3] = {256, 1024, 1024};
    const int boff[5] = {LRG_RT_B0, LRG_RT_B1, LRG_RT_B2, LRG_RT_B3, LRG_RT_B4}, bn[5] = {LRG_WB_C0, LRG_WB_C1, LRG_WB_C2, LRG_WB_C3, LRG_WB_C4};
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        if (l < 3) {
            const float4 *src = reinterpret_cast<const float4 *>(P.L[l].w);
            float4 *dst = reinterpret_cast<float4 *>(lrg_async_smem + ws + offs[l]);
            for (int i = tid; i < n4[l]; i += nthreads) dst[i] = src[i];
        }
        for (int i = tid; i < bn[l]; i += nthreads) lrg_async_smem[ws + boff[l] + i] = P.L[l].bias[i];
    }
}

// ---- the second kernel of a wave-branch launch (round 6): the CUs that run tiles, as a kernel of their own shape ----
// One launch used to give every role the front step's shape -- 1 024 threads, 128 VGPRs -- and the tile code inherited it (spills in the tile tasks, no room for a
// wavefront that keeps a tile's activations in registers: lrg_wave_tile.inl needs ~176).  This kernel is 512 threads (eight wavefronts, up to 256 VGPRs each) and is
// launched on a side stream right before lrg_grow_async_kernel, which then holds the front workgroups and the pooled-product units only; the two are resident together
// (one workgroup per CU in both: the front kernel fills its CUs' register files, this one's LDS allows no second workgroup on a CU; the host sizes both grids per
// XCD, tools/two_kernel_rendezvous.hip) and talk through the same rings and arrival counters as the roles of the single launch did.  Every workgroup of both kernels
// reports in; the front workgroups wait for all of them (bounded: reason 6).
//   workgroups 0 .. wave_a_wgs - 1: PREFIX CUs; wave_a_wgs .. wave_wgs - 1: POOL CUs of (side, half) = (w - wave_a_wgs) & 3 -- wavefronts 0 .. wave_waves - 1 run
//                                  branch tasks; wavefronts 4 .. 7 of the first fill_wgs are a fill-in team (wave_fill)
//   the others: two tile teams each on ring 1 (pooled blocks where there are no units, head tiles)
__global__ __launch_bounds__(LRG_WORKER_THREADS) void lrg_grow_async_worker_kernel(LrgAsyncKArgs K) {
    const int tid = threadIdx.x;
    const long long t_launch = wall_clock64();
#if defined(__HIP_DEVICE_COMPILE__)
    lrg_kargs_ptr kp = (lrg_kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
#else
    lrg_kargs_ptr kp = nullptr;
#endif
    if (tid == 0) __hip_atomic_fetch_add(&K.A.queue[LRG_AQ_ARRIVED], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int w = (int)blockIdx.x;
    if (w < K.A.wave_wgs) {
        const int rg = w < K.A.wave_a_wgs ? 4 : (w - K.A.wave_a_wgs) & 3;
        if (rg == 4) {
            lrg_wave_load_prefix_kernels(K.A.prob[0], 0, tid, LRG_WORKER_THREADS);
            lrg_wave_load_prefix_kernels(K.A.prob[1], LRG_WA_SIDE, tid, LRG_WORKER_THREADS);
        } else {
            lrg_wave_load_pool_kernels(K.A.prob[rg >> 1], 2 * (rg & 1), 0, tid, LRG_WORKER_THREADS);
            lrg_wave_load_pool_kernels(K.A.prob[rg >> 1], 2 * (rg & 1) + 1, LRG_WP_QUARTER, tid, LRG_WORKER_THREADS);
        }
        const bool fill_team = K.A.wave_fill && K.A.fill_list && w < K.A.fill_wgs && tid >= 256;
        int *word = reinterpret_cast<int *>(lrg_async_smem + LRG_WB_FLOATS);      // the fill-in team's region: control words first
        if (tid == 256) { word[4] = 0; word[5] = 0; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // the CU's kernels are in place; from here on every wavefront (and the fill-in team) goes its own way
        if (fill_team) { lrg_async_worker(kp, LRG_WB_FLOATS, t_launch, 1); return; }
        if ((tid >> 6) >= K.A.wave_waves) return;
        if (rg == 4) lrg_async_wave_prefix_worker(kp, t_launch);
        else lrg_async_wave_pool_worker(kp, rg, t_launch);
        return;
    }
    if (K.A.reg_tiles) {
        // a register-tile CU: [the kernels of layers 0 - 2 and the biases of both branches][team 0: control words, layer-3 exchange][team 1: a tile team's region]
        lrg_rt_load_kernels(K.A.prob[0], 0, tid, LRG_WORKER_THREADS);
        lrg_rt_load_kernels(K.A.prob[1], LRG_RT_SIDE, tid, LRG_WORKER_THREADS);
        const int sm_off = (tid >> 8) ? LRG_RT_WEIGHT_FLOATS + LRG_RT_TEAM0_FLOATS : LRG_RT_WEIGHT_FLOATS;
        int *word = reinterpret_cast<int *>(lrg_async_smem + sm_off);
        if ((tid & 255) == 0) { word[4] = 0; word[5] = 0; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!(tid >> 8)) lrg_async_reg_tile_team(kp, t_launch, LRG_RT_WEIGHT_FLOATS);
        else if (K.A.fill_list && w < K.A.fill_wgs) lrg_async_worker(kp, sm_off, t_launch, 1);      // (the fill-in ring on a few of them)
        else if (K.A.rt_bb_every > 0 && w % K.A.rt_bb_every == K.A.rt_bb_every - 1) lrg_async_reg_tile_team(kp, t_launch, sm_off);      // (a CU with two branch teams)
        else if (K.A.reg_tiles == 2) lrg_async_worker(kp, sm_off, t_launch, 2);                    // (LRG_ASYNC_RT_TEAM_HEADS=1: the head tiles as team tiles)
        else lrg_async_reg_head_team(kp, t_launch);
        return;
    }
    const int t = tid >> 8;
    if (t >= K.A.teams) return;
    const int sm_off = t * LRG_ASYNC_TEAM_FLOATS;
    int *word = reinterpret_cast<int *>(lrg_async_smem + sm_off);
    if ((tid & 255) == 0) { word[4] = 0; word[5] = 0; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    lrg_async_worker(kp, sm_off, t_launch, 2);
}
