// Probe for csrc/lrg_wave_tile.inl (round 6): one wavefront = one 32-row branch tile with the activations in registers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I learn_region_grow_amd/csrc tools/wave_tile_probe.hip -o gpurun_out/wave_tile_probe
// Checks conv[1] and the pooled maxima bit for bit against a host evaluation in the MFMA formulation's summation order (per output a chain of FMAs over
// k = 8g + 0, 4, 1, 5, 2, 6, 3, 7, then bias, then ReLU: lrg_fused_tile.inl), and times a task with 1 .. 16 wavefronts per CU on 1 / 64 / 200 CUs.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
extern __shared__ __attribute__((aligned(16))) float lrg_async_smem[];
#include "lrg_wave_tile.inl"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Args {
    const float *w[5];       // packed images (lrg_pack_weights layout)
    const float *b[5];
    const float *x, *center;
    float *conv1, *h3, *pool;     // pool: [slots][512]
    long long *cycles;       // [workgroups][16]
    int n_tiles, rows_per_slot, iters, waves, stage;
    LrgFusedProb head;       // stage 3: the register head tile's problem (x = conv[1] rows, L[0] / L[1], fw / fb / fout)
};

// stage 0: PREFIX tasks (layers 0 - 3 -> conv[1], layer-3 rows); stage 1: POOL tasks (workgroup & 1 = the half of the pooled layer's columns it holds)
__global__ __launch_bounds__(512) void probe_kernel(Args a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = blockIdx.x & 1;
    if (a.stage == 0) {
        const int sizes[4] = {1024, 4096, 4096, 8192}, offs[4] = {LRG_WA_W0, LRG_WA_W1, LRG_WA_W2, LRG_WA_W3};
        const int bs[4] = {64, 64, 64, 128}, bo[4] = {LRG_WA_B0, LRG_WA_B1, LRG_WA_B2, LRG_WA_B3};
        for (int side = 0; side < 2; ++side)      // (the probe has one branch: both halves of the LDS hold it, the odd tiles use the second copy)
            for (int l = 0; l < 4; ++l) {
                for (int i = tid; i < sizes[l] / 4; i += 512) reinterpret_cast<float4 *>(lrg_async_smem + side * LRG_WA_SIDE + offs[l])[i] = reinterpret_cast<const float4 *>(a.w[l])[i];
                for (int i = tid; i < bs[l]; i += 512) lrg_async_smem[side * LRG_WA_SIDE + bo[l] + i] = a.b[l][i];
            }
    } else {
        for (int qq = 0; qq < 2; ++qq) {
            const int q = 2 * half + qq;
            for (int i = tid; i < 4096; i += 512) reinterpret_cast<float4 *>(lrg_async_smem + qq * LRG_WP_QUARTER + LRG_WP_W4)[i] = reinterpret_cast<const float4 *>(a.w[4] + (long)q * 16384)[i];
            for (int i = tid; i < 128; i += 512) lrg_async_smem[qq * LRG_WP_QUARTER + LRG_WP_B4 + i] = a.b[4][q * 128 + i];
        }
    }
    if (a.stage == 3) {
        // REGISTER HEAD TILE: one team of four wavefronts per tile
        LrgWgTeam team;
        struct { __device__ void operator()() const {} } nowait;
        const long long t0 = (long long)__builtin_readcyclecounter();
        for (int it = 0; it < a.iters; ++it) {
            const int tile = (blockIdx.x + it * 7) % a.n_tiles;
            const long r0 = (long)tile * 32;
            lrg_team_head_tile_reg(a.head, r0, (int)(r0 / a.rows_per_slot), 0, team, nowait, wave, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const long long t1 = (long long)__builtin_readcyclecounter();
        if (lane == 0) a.cycles[blockIdx.x * 16 + wave] = t1 - t0;
        return;
    }
    if (a.stage == 2) {
        // REGISTER TILE: the workgroup = one team of four wavefronts per tile; both halves of the kernels' LDS hold the probe's one branch
        const int sizes[3] = {1024, 4096, 4096}, offs[3] = {LRG_RT_W0, LRG_RT_W1, LRG_RT_W2};
        const int bs[5] = {64, 64, 64, 128, 512}, bo[5] = {LRG_RT_B0, LRG_RT_B1, LRG_RT_B2, LRG_RT_B3, LRG_RT_B4};
        for (int side = 0; side < 2; ++side) {
            for (int l = 0; l < 3; ++l)
                for (int i = tid; i < sizes[l] / 4; i += blockDim.x) reinterpret_cast<float4 *>(lrg_async_smem + side * LRG_RT_SIDE + offs[l])[i] = reinterpret_cast<const float4 *>(a.w[l])[i];
            for (int l = 0; l < 5; ++l)
                for (int i = tid; i < bs[l]; i += blockDim.x) lrg_async_smem[side * LRG_RT_SIDE + bo[l] + i] = a.b[l][i];
        }
        __syncthreads();
        LrgWgTeam team;
        const long long t0 = (long long)__builtin_readcyclecounter();
        for (int it = 0; it < a.iters; ++it) {
            const int tile = (blockIdx.x + it * 7) % a.n_tiles;
            const long r0 = (long)tile * 32;
            lrg_team_branch_tile_reg<1>(a.x, a.center, a.conv1, a.pool + (r0 / a.rows_per_slot) * 512, a.w[3], a.w[4], r0, (int)(r0 / a.rows_per_slot), (tile & 1) * LRG_RT_SIDE,
                                     LRG_RT_WEIGHT_FLOATS, team, wave, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                  // (as the task does before its arrival; the exchange buffer is free again)
        }
        const long long t1 = (long long)__builtin_readcyclecounter();
        if (lane == 0) a.cycles[blockIdx.x * 16 + wave] = t1 - t0;
        return;
    }
    __syncthreads();
    if (wave >= a.waves) return;
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int it = 0; it < a.iters; ++it) {
        const int job = (blockIdx.x >> 1) * a.waves + wave + it * 7;
        if (a.stage == 0) {
            const int tile = (2 * job + half) % a.n_tiles;
            const long r0 = (long)tile * 32;
            lrg_wave_prefix_tile(a.x, a.center, a.conv1, a.h3, r0, (int)(r0 / a.rows_per_slot), (tile & 1) * LRG_WA_SIDE, lane);
        } else {
            const int tile = (job >> 1) % a.n_tiles, q = 2 * half + (job & 1);
            const long r0 = (long)tile * 32;
            lrg_wave_pool_tile(a.h3, a.pool + (r0 / a.rows_per_slot) * 512, r0, q, (q & 1) * LRG_WP_QUARTER, 0, 2, lane);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (lane == 0) a.cycles[blockIdx.x * 16 + wave] = t1 - t0;
}

static void pack(const std::vector<float> &w, int K, int N, std::vector<float> &out) {
    const int ng = (K + 7) / 8, ncb = (N + 31) / 32;
    out.assign((size_t)ncb * ng * 64 * 4, 0.f);
    for (int cb = 0; cb < ncb; ++cb)
        for (int g = 0; g < ng; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 4; ++s) {
                    const int col = 32 * cb + (lane & 31), k = 8 * g + 4 * (lane >> 5) + s;
                    out[(((size_t)cb * ng + g) * 64 + lane) * 4 + s] = (k < K && col < N) ? w[(size_t)k * N + col] : 0.f;
                }
}

static void layer_ref(const std::vector<float> &in, int rows, int K, const std::vector<float> &w, const std::vector<float> &b, int N, std::vector<float> &out) {
    const int ng = (K + 7) / 8;
    out.assign((size_t)rows * N, 0.f);
    for (int n = 0; n < rows; ++n)
        for (int c = 0; c < N; ++c) {
            float acc = 0.f;
            for (int g = 0; g < ng; ++g)
                for (int s = 0; s < 4; ++s)
                    for (int h = 0; h < 2; ++h) {
                        const int k = 8 * g + 4 * h + s;
                        const float xv = k < K ? in[(size_t)n * K + k] : 0.f, wv = k < K ? w[(size_t)k * N + c] : 0.f;
                        acc = fmaf(wv, xv, acc);
                    }
            out[(size_t)n * N + c] = fmaxf(acc + b[c], 0.f);
        }
}

int main(int argc, char **argv) {
    const int n_slots = 6, rows_per_slot = 96, rows = n_slots * rows_per_slot, n_tiles = rows / 32, F = 13;
    const int Ks[5] = {F, 64, 64, 64, 128}, Ns[5] = {64, 64, 64, 128, 512};
    srand(12345);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    std::vector<float> w[5], b[5], pk[5];
    for (int l = 0; l < 5; ++l) {
        w[l].resize((size_t)Ks[l] * Ns[l]); b[l].resize(Ns[l]);
        const float sc = 1.5f / sqrtf((float)Ks[l]);
        for (auto &v : w[l]) v = rnd() * sc;
        for (auto &v : b[l]) v = rnd() * 0.1f;
        pack(w[l], Ks[l], Ns[l], pk[l]);
    }
    std::vector<float> x16((size_t)rows * 16, 0.f), cen((size_t)n_slots * 16, 0.f), xc((size_t)rows * F);
    for (int s = 0; s < n_slots; ++s)
        for (int c = 0; c < F; ++c) cen[s * 16 + c] = (c < 2 || c >= 6) ? rnd() : 0.f;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < F; ++c) {
            x16[(size_t)r * 16 + c] = rnd() * 2.f;
            xc[(size_t)r * F + c] = x16[(size_t)r * 16 + c] - cen[(r / rows_per_slot) * 16 + c];
        }
    // host reference
    std::vector<float> h[5];
    layer_ref(xc, rows, F, w[0], b[0], 64, h[0]);
    for (int l = 1; l < 5; ++l) layer_ref(h[l - 1], rows, Ks[l], w[l], b[l], Ns[l], h[l]);
    std::vector<float> pool_ref((size_t)n_slots * 512, 0.f);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < 512; ++c) pool_ref[(size_t)(r / rows_per_slot) * 512 + c] = fmaxf(pool_ref[(size_t)(r / rows_per_slot) * 512 + c], h[4][(size_t)r * 512 + c]);

    Args a;
    float *d;
    for (int l = 0; l < 5; ++l) {
        CK(hipMalloc(&d, pk[l].size() * 4)); CK(hipMemcpy(d, pk[l].data(), pk[l].size() * 4, hipMemcpyHostToDevice)); a.w[l] = d;
        CK(hipMalloc(&d, b[l].size() * 4)); CK(hipMemcpy(d, b[l].data(), b[l].size() * 4, hipMemcpyHostToDevice)); a.b[l] = d;
    }
    CK(hipMalloc(&d, x16.size() * 4)); CK(hipMemcpy(d, x16.data(), x16.size() * 4, hipMemcpyHostToDevice)); a.x = d;
    CK(hipMalloc(&d, cen.size() * 4)); CK(hipMemcpy(d, cen.data(), cen.size() * 4, hipMemcpyHostToDevice)); a.center = d;
    float *conv1, *pool, *h3;
    CK(hipMalloc(&conv1, (size_t)rows * 64 * 4)); CK(hipMemset(conv1, 0xFF, (size_t)rows * 64 * 4));
    CK(hipMalloc(&h3, (size_t)rows * 128 * 4)); CK(hipMemset(h3, 0xFF, (size_t)rows * 128 * 4));
    CK(hipMalloc(&pool, (size_t)n_slots * 512 * 4)); CK(hipMemset(pool, 0, (size_t)n_slots * 512 * 4));
    long long *cyc;
    CK(hipMalloc(&cyc, 256 * 16 * 8));
    a.conv1 = conv1; a.h3 = h3; a.pool = pool; a.cycles = cyc; a.n_tiles = n_tiles; a.rows_per_slot = rows_per_slot;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = LRG_WB_FLOATS * 4;

    // ---- correctness: every tile's PREFIX task, then every (tile, quarter) POOL task, at least once ----
    a.iters = 1; a.waves = 4;
    const int wgs = 2 * ((n_tiles + 3) / 4) * 2;      // (each tile several times: the stores and the maxima are idempotent)
    a.stage = 0;
    hipLaunchKernelGGL(probe_kernel, dim3(wgs), dim3(512), lds, 0, a);
    CK(hipDeviceSynchronize());
    a.stage = 1;
    hipLaunchKernelGGL(probe_kernel, dim3(2 * wgs), dim3(512), lds, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> g1((size_t)rows * 64), g3((size_t)rows * 128), gp((size_t)n_slots * 512);
    CK(hipMemcpy(g1.data(), conv1, g1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(g3.data(), h3, g3.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gp.data(), pool, gp.size() * 4, hipMemcpyDeviceToHost));
    size_t bad1 = 0, bad3 = 0, badp = 0;
    for (size_t i = 0; i < g1.size(); ++i) bad1 += memcmp(&g1[i], &h[1][i], 4) != 0;
    for (size_t i = 0; i < g3.size(); ++i) bad3 += memcmp(&g3[i], &h[3][i], 4) != 0;
    for (size_t i = 0; i < gp.size(); ++i) badp += memcmp(&gp[i], &pool_ref[i], 4) != 0;
    printf("conv[1]: %zu of %zu values differ from the host chain; layer 3: %zu of %zu; pooled: %zu of %zu\n", bad1, g1.size(), bad3, g3.size(), badp, gp.size());
    if (bad1 || bad3 || badp) {
        for (size_t i = 0, shown = 0; i < g1.size() && shown < 8; ++i)
            if (memcmp(&g1[i], &h[1][i], 4)) { printf("  conv1[%zu,%zu] gpu %.9g ref %.9g\n", i / 64, i % 64, g1[i], h[1][i]); ++shown; }
        for (size_t i = 0, shown = 0; i < gp.size() && shown < 8; ++i)
            if (memcmp(&gp[i], &pool_ref[i], 4)) { printf("  pool[%zu,%zu] gpu %.9g ref %.9g\n", i / 512, i % 512, gp[i], pool_ref[i]); ++shown; }
    }
    badp += bad3;
    // ---- the register tile (a team of four wavefronts per tile) ----
    CK(hipMemset(conv1, 0xFF, (size_t)rows * 64 * 4)); CK(hipMemset(pool, 0, (size_t)n_slots * 512 * 4));
    a.stage = 2; a.iters = 1;
    hipLaunchKernelGGL(probe_kernel, dim3(n_tiles), dim3(256), (LRG_RT_WEIGHT_FLOATS + LRG_RT_XCH_FLOATS) * 4, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(g1.data(), conv1, g1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gp.data(), pool, gp.size() * 4, hipMemcpyDeviceToHost));
    size_t rb1 = 0, rbp = 0;
    for (size_t i = 0; i < g1.size(); ++i) rb1 += memcmp(&g1[i], &h[1][i], 4) != 0;
    for (size_t i = 0; i < gp.size(); ++i) rbp += memcmp(&gp[i], &pool_ref[i], 4) != 0;
    printf("register tile (team of four): conv[1] %zu of %zu differ; pooled %zu of %zu\n", rb1, g1.size(), rbp, gp.size());
    badp += rb1 + rbp;
    for (int grid : {1, 200}) {
        a.iters = 50;
        hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(256), (LRG_RT_WEIGHT_FLOATS + LRG_RT_XCH_FLOATS) * 4, 0, a);
        CK(hipDeviceSynchronize());
        std::vector<long long> c((size_t)grid * 16);
        CK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (int g = 0; g < grid; ++g) sum += (double)c[g * 16] / a.iters;
        printf("REGISTER TILE, grid %3d, one team per CU: %8.0f counter ticks per tile (the team tile of lrg_fused_tile.inl: ~52 000)\n", grid, sum / grid);
    }

    // ---- timing (counter ticks of the shader clock; a PREFIX task is 272 x 64 = 17 408 cycles of MFMA issue, a POOL task 256 x 64 = 16 384) ----
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const double ghz = prop.clockRate * 1e-6;
    for (int stage = 0; stage < 2; ++stage)
        for (int grid : {2, 200}) {
            for (int waves : {1, 4, 8}) {
                a.iters = 50; a.waves = waves; a.stage = stage;
                hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(512), lds, 0, a);
                CK(hipDeviceSynchronize());
                std::vector<long long> c((size_t)grid * 16);
                CK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
                double sum = 0, mx = 0;
                for (int g = 0; g < grid; ++g)
                    for (int wv = 0; wv < waves; ++wv) { const double v = (double)c[g * 16 + wv] / a.iters; sum += v; mx = v > mx ? v : mx; }
                printf("%s, grid %3d, %2d waves per CU: %8.0f counter ticks per task (max %8.0f)\n", stage ? "POOL  " : "PREFIX", grid, waves, sum / (grid * waves), mx);
            }
        }
    // ---- the register head tile: conv[1] rows (the branch reference's) + a per-slot pooled product -> 256 -> 128 -> 2 ----
    {
        std::vector<float> hw0((size_t)64 * 256), hw1((size_t)256 * 128), hbias1(128), fw(256), fb(2), hbv((size_t)n_slots * 256), pk0, pk1;
        for (auto &v : hw0) v = rnd() * 0.2f;
        for (auto &v : hw1) v = rnd() * 0.1f;
        for (auto &v : hbias1) v = rnd() * 0.1f;
        for (auto &v : fw) v = rnd() * 0.2f;
        for (auto &v : fb) v = rnd() * 0.1f;
        for (auto &v : hbv) v = rnd() * 0.5f;
        pack(hw0, 64, 256, pk0); pack(hw1, 256, 128, pk1);
        // host: the MFMA chain per layer, then lrg_fused_tile's last layer (eight partial chains over k = 4q + 32m + i, butterfly over q)
        std::vector<float> g0((size_t)rows * 256), g1h((size_t)rows * 128), lref((size_t)rows * 2);
        const std::vector<float> &cin = h[1];
        for (int n = 0; n < rows; ++n) {
            for (int c = 0; c < 256; ++c) {
                float acc = 0.f;
                for (int g = 0; g < 8; ++g) for (int s2 = 0; s2 < 4; ++s2) for (int hh = 0; hh < 2; ++hh) { const int k = 8 * g + 4 * hh + s2; acc = fmaf(hw0[(size_t)k * 256 + c], cin[(size_t)n * 64 + k], acc); }
                g0[(size_t)n * 256 + c] = fmaxf((acc + hbv[(size_t)(n / rows_per_slot) * 256 + c]) + 0.f, 0.f);
            }
            for (int c = 0; c < 128; ++c) {
                float acc = 0.f;
                for (int g = 0; g < 32; ++g) for (int s2 = 0; s2 < 4; ++s2) for (int hh = 0; hh < 2; ++hh) { const int k = 8 * g + 4 * hh + s2; acc = fmaf(hw1[(size_t)k * 128 + c], g0[(size_t)n * 256 + k], acc); }
                g1h[(size_t)n * 128 + c] = fmaxf(acc + hbias1[c], 0.f);
            }
            float part[8][2];
            for (int q = 0; q < 8; ++q) {
                float s0 = 0.f, s1 = 0.f;
                for (int k = 4 * q; k < 128; k += 32)
                    for (int i = 0; i < 4; ++i) { s0 = fmaf(g1h[(size_t)n * 128 + k + i], fw[2 * (k + i)], s0); s1 = fmaf(g1h[(size_t)n * 128 + k + i], fw[2 * (k + i) + 1], s1); }
                part[q][0] = s0; part[q][1] = s1;
            }
            for (int o = 0; o < 2; ++o) {
                const float t01 = part[0][o] + part[1][o], t23 = part[2][o] + part[3][o], t45 = part[4][o] + part[5][o], t67 = part[6][o] + part[7][o];
                lref[(size_t)n * 2 + o] = ((t01 + t23) + (t45 + t67)) + fb[o];
            }
        }
        auto up = [&](const std::vector<float> &v) { float *dd; CK(hipMalloc(&dd, v.size() * 4)); CK(hipMemcpy(dd, v.data(), v.size() * 4, hipMemcpyHostToDevice)); return dd; };
        LrgFusedProb &H = a.head;
        memset(&H, 0, sizeof(H));
        H.x = up(cin); H.L[0].w = up(pk0); H.L[0].bias = up(hbv); H.L[1].w = up(pk1); H.L[1].bias = up(hbias1); H.fw = up(fw); H.fb = up(fb);
        float *lout; CK(hipMalloc(&lout, (size_t)rows * 2 * 4)); CK(hipMemset(lout, 0xFF, (size_t)rows * 2 * 4)); H.fout = lout;
        a.stage = 3; a.iters = 1;
        hipLaunchKernelGGL(probe_kernel, dim3(n_tiles), dim3(256), LRG_RH_FLOATS * 4, 0, a);
        CK(hipDeviceSynchronize());
        std::vector<float> lg((size_t)rows * 2);
        CK(hipMemcpy(lg.data(), lout, lg.size() * 4, hipMemcpyDeviceToHost));
        size_t bl = 0;
        for (size_t i = 0; i < lg.size(); ++i) bl += memcmp(&lg[i], &lref[i], 4) != 0;
        printf("register head tile: logits %zu of %zu differ from the host chain\n", bl, lg.size());
        for (size_t i = 0, shown = 0; i < lg.size() && shown < 6; ++i) if (memcmp(&lg[i], &lref[i], 4)) { printf("  logit[%zu,%zu] gpu %.9g ref %.9g\n", i / 2, i % 2, lg[i], lref[i]); ++shown; }
        badp += bl;
        for (int grid : {1, 200}) {
            a.iters = 50;
            hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(256), LRG_RH_FLOATS * 4, 0, a);
            CK(hipDeviceSynchronize());
            std::vector<long long> c((size_t)grid * 16);
            CK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
            double sum = 0;
            for (int g = 0; g < grid; ++g) sum += (double)c[g * 16] / a.iters;
            printf("REGISTER HEAD TILE, grid %3d, one team per CU: %8.0f counter ticks per tile\n", grid, sum / grid);
        }
    }
    (void)ghz;
    (void)argc; (void)argv;
    return (bad1 || badp) ? 1 : 0;
}
