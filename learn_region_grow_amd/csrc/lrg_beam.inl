// Beam-search region growing (test_beam_search.py:143-290) with the queue ON THE DEVICE; included by lrg_grow.hip.
//
// Per seed the reference keeps a queue Q of at most BEAM_WIDTH masks; every entry spawns SEARCH_WIDTH stochastic grow steps;
// the children whose mask changed are scored by size (--scoring np), the best BEAM_WIDTH become the next queue; the queue's
// head is the answer when its bounding box has not grown twice in a row (:188-198) or no child survives (:179).  Here a room
// in flight is a GROUP of G = BEAM_WIDTH * SEARCH_WIDTH slots -- child (qid, search id) is slot qid * SEARCH_WIDTH + search id
// -- and one level of every room in flight is one pass of the loop's kernels over all slots, bracketed by lrg_beam_advance:
// one workgroup per group that (1) turns the children's results into the next queue (stable order by size, masks copied into
// the group's parent buffers), (2) applies the stall test / commits the head / picks the next seed, (3) writes the next
// children into the slots.  Nothing returns to the host between levels; finished rooms are announced through the stats ring
// as in greedy growing.  A child's random stream is keyed (seed point, child ordinal, level), as oracle/beam_ref.py keys it.
#define LRG_BEAM_MAXQ 16      // BEAM_WIDTH <= 16, BEAM_WIDTH * SEARCH_WIDTH <= 64

__device__ void lrg_beam_commit(LrgBeamGroup *B, LrgRoom *R, const LrgGrowParams &prm, int64_t *stats) {
    // visited / label from the head of the queue (:289-293); workgroup-wide, ends with a barrier
    const int n = R->n;
    const int parent = B->q_parent[0];
    const int count = B->q_count[0];
    const int labeled = count > prm.cluster_threshold;
    const int cid = R->next_cluster_id;
    if (parent < 0) {
        if (threadIdx.x == 0) { R->visited[B->seed] = 1; if (labeled) R->label[B->seed] = cid; }
    } else {
        const uint8_t *mask = B->parent + (long)parent * B->cap;
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if (mask[i]) { R->visited[i] = 1; if (labeled) R->label[i] = cid; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t *log = R->region_log + LRG_LOG_WORDS * (long)R->n_regions;
        log[0] = B->seed; log[1] = B->steps; log[2] = count; log[3] = 0; log[4] = labeled; log[5] = 0; log[6] = -1; log[7] = -1;
        R->n_regions += 1;
        if (labeled) R->next_cluster_id = cid + 1;
        if (stats) atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), 1ULL);
        B->seed = -1;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void lrg_beam_advance_kernel(LrgBeamGroup *groups, LrgSlot *slots, LrgRoom *rooms, int BW, int SW,
                                                                 LrgGrowParams prm, int64_t *stats) {
    __shared__ int sh_next, sh_sel[LRG_BEAM_MAXQ], sh_nsel;
    const int g = blockIdx.x, tid = threadIdx.x, G = BW * SW;
    LrgBeamGroup *B = &groups[g];
    LrgSlot *S0 = &slots[g * G];
    if (B->room < 0 || B->done) {
        if (tid < G) { S0[tid].room = -1; S0[tid].status = LRG_IDLE; }
        return;
    }
    LrgRoom *R = &rooms[B->room];
    const int n = R->n;
    // ---- (1) the level just evaluated -> the next queue (:275-287) ----
    if (B->pending) {
        if (tid == 0) {
            const int nq = B->nq;
            int ran = 0;
            for (int q = 0; q < nq; ++q) ran += S0[q * SW].updated >= 0;          // a parent without neighbours spawns nothing (:212)
            B->steps += SW * ran;                                                  // :274, one per child
            // children whose mask changed and is not empty, by size, ties in (qid, search id) order (a stable sort, :286)
            int ns = 0;
            for (int c = 0; c < nq * SW; ++c) {
                const LrgSlot *S = &S0[c];
                if (S->updated == 1 && S->scan_cnt > 0) {
                    int k = ns < BW ? ns : BW;                                     // insertion into the best-BW list
                    while (k > 0 && S0[sh_sel[k - 1]].scan_cnt < S->scan_cnt) { if (k < BW) sh_sel[k] = sh_sel[k - 1]; --k; }
                    if (k < BW) { sh_sel[k] = c; if (ns < BW) ++ns; }
                }
            }
            sh_nsel = ns;
        }
        __syncthreads();
        const int ns = sh_nsel;
        if (ns == 0) {
            lrg_beam_commit(B, R, prm, stats);             // the queue ran dry: its last head is the answer (:179,:289)
        } else {
            for (int k = 0; k < ns; ++k) {                  // the survivors' masks become the parents of the next level
                const uint8_t *src = S0[sh_sel[k]].cur;
                uint8_t *dst = B->parent + (long)k * B->cap;
                for (int i = tid; i < n; i += blockDim.x) dst[i] = src[i];
            }
            __syncthreads();
            if (tid == 0) {
                for (int k = 0; k < ns; ++k) {
                    const LrgSlot *S = &S0[sh_sel[k]];
                    B->q_count[k] = S->scan_cnt; B->q_parent[k] = k;
                    for (int d = 0; d < 3; ++d) { B->q_mn[k][d] = S->scan_mn[d]; B->q_mx[k][d] = S->scan_mx[d]; }
                }
                B->nq = ns;
                B->level += 1;
            }
            __syncthreads();
        }
        if (tid == 0) B->pending = 0;
        __syncthreads();
    }
    // ---- (2) stall test on the head (:188-198), commit, next seed (:154-174) ----
    for (;;) {
        if (B->seed < 0) {
            int cursor = R->seed_cursor, found = -1;
            while (cursor < n) {
                const int pos = cursor + tid;
                int cand = INT_MAX;
                if (pos < n && !R->visited[R->order[pos]]) cand = pos;
                if (tid == 0) sh_next = INT_MAX;
                __syncthreads();
                if (cand != INT_MAX) atomicMin(&sh_next, cand);
                __syncthreads();
                const int best = sh_next;
                __syncthreads();
                if (best != INT_MAX) { found = best; break; }
                cursor += blockDim.x;
            }
            if (found < 0) {
                if (tid == 0) {
                    R->seed_cursor = n; R->done = 1; B->done = 1;
                    if (stats) {
                        unsigned long long k = atomicAdd(reinterpret_cast<unsigned long long *>(&stats[1]), 1ULL);
                        stats[4 + (k % LRG_DONE_RING)] = g * G;
                    }
                }
                if (tid < G) { S0[tid].room = -1; S0[tid].status = LRG_IDLE; }
                return;
            }
            if (tid == 0) {
                const int sd = R->order[found];
                R->seed_cursor = found + 1;
                B->seed = sd; B->level = 0; B->stuck = 0; B->steps = 0; B->nq = 1;
                B->q_count[0] = 1; B->q_parent[0] = -1;                          // the seed-only mask (:164-174)
                for (int d = 0; d < 3; ++d) {
                    const int v = R->voxels[3 * sd + d];
                    B->q_mn[0][d] = v; B->q_mx[0][d] = v; B->seq_mn[d] = v; B->seq_mx[d] = v;
                }
            }
            __syncthreads();
        }
        bool commit = false;
        if (tid == 0) {
            bool grew = false;
            for (int d = 0; d < 3; ++d) grew |= B->q_mn[0][d] < B->seq_mn[d] || B->q_mx[0][d] > B->seq_mx[d];
            if (!grew) {
                if (B->stuck >= 1) sh_next = 1; else { B->stuck += 1; sh_next = 0; }
            } else { B->stuck = 0; sh_next = 0; }
            if (!sh_next)
                for (int d = 0; d < 3; ++d) { B->seq_mn[d] = min(B->seq_mn[d], B->q_mn[0][d]); B->seq_mx[d] = max(B->seq_mx[d], B->q_mx[0][d]); }
        }
        __syncthreads();
        commit = sh_next != 0;
        __syncthreads();
        if (!commit) break;
        lrg_beam_commit(B, R, prm, stats);
    }
    // ---- (3) the children of this level ----
    const int nq = B->nq, seed = B->seed;
    for (int c = 0; c < nq * SW; ++c) {
        const int q = c / SW;
        uint8_t *dst = S0[c].cur;
        const int parent = B->q_parent[q];
        if (parent < 0) { for (int i = tid; i < n; i += blockDim.x) dst[i] = i == seed; }
        else {
            const uint8_t *src = B->parent + (long)parent * B->cap;
            for (int i = tid; i < n; i += blockDim.x) dst[i] = src[i];
        }
    }
    if (tid < G) {
        LrgSlot *S = &S0[tid];
        if (tid < nq * SW) {
            const int q = tid / SW;
            S->room = B->room; S->status = LRG_ACTIVE; S->seed = seed; S->step = B->level; S->restart = tid;   // RNG key: (seed, ordinal, level)
            S->steps_total = 0; S->stuck = 0; S->pad = 0; S->nc = 0; S->ne = 0; S->query = 0; S->scan_cnt = 0; S->updated = -1;
            S->count = B->q_count[q];
            S->target = R->obj_id ? R->obj_id[seed] : 0;
            for (int d = 0; d < 3; ++d) {
                S->mn[d] = B->q_mn[q][d]; S->mx[d] = B->q_mx[q][d];
                S->scan_mn[d] = INT_MAX; S->scan_mx[d] = INT_MIN;
            }
        } else { S->room = -1; S->status = LRG_IDLE; }
    }
    if (tid == 0) B->pending = 1;
}
