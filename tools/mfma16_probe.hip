// Probe (not part of the library) for v_mfma_f32_16x16x4_f32 as the instruction of a 32-row LrgNet tile:
//   (1) in which order it adds its four products (compared with float32 FMA chains emulated on the host -- the tiles' results must
//       stay bit for bit those of the v_mfma_f32_32x32x2_f32 formulation, whose k-groups of 8 are added as k = 0, 4, 1, 5, 2, 6, 3, 7);
//   (2) its sustained rate for one four-wavefront team per CU with the operand streams of the fused tile (A from LDS, B from a
//       register ring fed from L2), four independent 16x16 accumulators per wavefront, next to the 32x32x2 single chain.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma16_probe tools/mfma16_probe.hip && /tmp/mfma16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- (1) order ----
__global__ void order_kernel(const float *A, const float *B, const float *C, float *D, int chain2) {
    // one wavefront per problem: A [16][4], B [4][16], C [16][16] row-major; chain2: a second instruction on k = 4..7 of A [16][8], B [8][16]
    const int lane = threadIdx.x, p = blockIdx.x;
    const int K = chain2 ? 8 : 4;
    const float *a = A + p * 16 * K, *b = B + p * K * 16, *c = C + p * 256;
    float *d = D + p * 256;
    const int i = lane & 15, q = lane >> 4;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = c[(4 * q + r) * 16 + i];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i * K + q], b[q * 16 + i], acc, 0, 0, 0);
    if (chain2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i * K + 4 + q], b[(4 + q) * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[(4 * q + r) * 16 + i] = acc[r];
}

static void order_check() {
    const int NP = 512;
    std::vector<float> A(NP * 16 * 8), B(NP * 8 * 16), C(NP * 256), D(NP * 256);
    srand(1);
    auto rnd = [] { return (float)((rand() / (double)RAND_MAX - 0.5) * 4.0); };
    for (auto &v : A) v = rnd();
    for (auto &v : B) v = rnd();
    for (auto &v : C) v = rnd() * 3.f;
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    for (int chain2 = 0; chain2 < 2; ++chain2) {
        const int K = chain2 ? 8 : 4;
        // repack A, B for this K (the kernel indexes [16][K], [K][16])
        std::vector<float> a(NP * 16 * K), b(NP * K * 16);
        for (int p = 0; p < NP; ++p) {
            for (int i = 0; i < 16; ++i) for (int k = 0; k < K; ++k) a[(p * 16 + i) * K + k] = A[(p * 16 + i) * 8 + k];
            for (int k = 0; k < K; ++k) for (int j = 0; j < 16; ++j) b[(p * K + k) * 16 + j] = B[(p * 8 + k) * 16 + j];
        }
        hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(order_kernel, dim3(NP), dim3(64), 0, 0, dA, dB, dC, dD, chain2);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        // hypotheses: FMA chain over k in every order of the first four (then 4..7 in the same order), and the exact sum rounded once
        int perm[24][4], np = 0;
        for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) for (int z = 0; z < 4; ++z) for (int w = 0; w < 4; ++w)
            if (x != y && x != z && x != w && y != z && y != w && z != w) { perm[np][0] = x; perm[np][1] = y; perm[np][2] = z; perm[np][3] = w; ++np; }
        long match[25] = {0}, total = 0;
        for (int p = 0; p < NP; ++p) for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            const float got = D[p * 256 + i * 16 + j];
            for (int h = 0; h < 24; ++h) {
                float acc = C[p * 256 + i * 16 + j];
                for (int part = 0; part < (chain2 ? 2 : 1); ++part)
                    for (int t = 0; t < 4; ++t) { const int k = 4 * part + perm[h][t]; acc = fmaf(a[(p * 16 + i) * K + k], b[(p * K + k) * 16 + j], acc); }
                match[h] += acc == got;
            }
            double ex = C[p * 256 + i * 16 + j];
            float accp = C[p * 256 + i * 16 + j];
            for (int part = 0; part < (chain2 ? 2 : 1); ++part) {
                double s = accp;
                for (int t = 0; t < 4; ++t) { const int k = 4 * part + t; s += (double)a[(p * 16 + i) * K + k] * (double)b[(p * K + k) * 16 + j]; }
                accp = (float)s;
            }
            (void)ex;
            match[24] += accp == got;
            ++total;
        }
        printf("order check, %d instruction(s) per accumulator: %ld outputs\n", chain2 ? 2 : 1, total);
        for (int h = 0; h < 24; ++h)
            if (match[h] > total / 2 || h == 0)
                printf("  FMA chain k = %d %d %d %d : %.4f\n", perm[h][0], perm[h][1], perm[h][2], perm[h][3], match[h] / (double)total);
        printf("  exact sum of four, rounded once per instruction : %.4f\n", match[24] / (double)total);
    }
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
}

// ---- (2) rate ----
// MODE bit0: A from LDS; bit1: B through a ring of FD k-groups of 8 (one dwordx4 per lane per group) that never drains.
// V = 0: v_mfma_f32_32x32x2_f32, one accumulator chain (the tile today); V = 1: v_mfma_f32_16x16x4_f32, 2 x 2 blocks = four chains,
// A as one ds_read_b64 per row half per group (columns permuted within a group so that a lane's two k values are neighbours);
// V = 2: the same with A as one ds_read_b128 per row half per TWO groups.
template <int MODE, int V, int FD>
__global__ __launch_bounds__(256, 1) void rate_kernel(const float *w, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    for (int i = tid; i < 64 * 132; i += 256) smem[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    float4 bq[FD];
    auto loadb = [&](int it, int g) -> float4 {      // packed: [pass][wave][group][lane][4]
        return *reinterpret_cast<const float4 *>(w + ((((it & 3) * 4 + wn) * 16 + g) * 64 + lane) * 4);
    };
    for (int g = 0; g < FD; ++g) bq[g] = (MODE & 2) ? loadb(0, g) : make_float4(1.f, 2.f, 0.5f, 0.25f);
    float s = 0.f;
    if (V == 0) {
        const int li = lane & 31, lh = lane >> 5;
        const float *ap = smem + li * 132 + 4 * lh;
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float4 a = make_float4(1.f, 2.f, 3.f, 4.f);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (MODE & 1) a = *reinterpret_cast<const float4 *>(ap + 8 * g);
                const float4 b = bq[g % FD];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
                if (MODE & 2) bq[g % FD] = (g + FD < 16) ? loadb(it, g + FD) : loadb(it + 1, g + FD - 16);
            }
        }
        for (int i = 0; i < 16; ++i) s += acc[i];
    } else {
        const int i16 = lane & 15, q = lane >> 4;
        f32x4 acc[2][2];
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) for (int i = 0; i < 4; ++i) acc[r][c][i] = 0.f;
        float2 a2[2] = {make_float2(1.f, 2.f), make_float2(3.f, 4.f)};
        float4 a4[2] = {make_float4(1.f, 2.f, 3.f, 4.f), make_float4(1.f, 2.f, 3.f, 4.f)};
        const float *ap = smem + i16 * 132 + (V == 1 ? 2 * q : 4 * q);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                float ax[2], ay[2];
                if (V == 1) {
                    if (MODE & 1) { a2[0] = *reinterpret_cast<const float2 *>(ap + 8 * g); a2[1] = *reinterpret_cast<const float2 *>(ap + 16 * 132 + 8 * g); }
                    ax[0] = a2[0].x; ay[0] = a2[0].y; ax[1] = a2[1].x; ay[1] = a2[1].y;
                } else {
                    if ((MODE & 1) && !(g & 1)) { a4[0] = *reinterpret_cast<const float4 *>(ap + 8 * g); a4[1] = *reinterpret_cast<const float4 *>(ap + 16 * 132 + 8 * g); }
                    ax[0] = (g & 1) ? a4[0].z : a4[0].x; ay[0] = (g & 1) ? a4[0].w : a4[0].y;
                    ax[1] = (g & 1) ? a4[1].z : a4[1].x; ay[1] = (g & 1) ? a4[1].w : a4[1].y;
                }
                const float4 b = bq[g % FD];
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[0], b.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[0], b.z, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[1], b.x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[1], b.z, acc[1][1], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ay[0], b.y, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ay[0], b.w, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ay[1], b.y, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ay[1], b.w, acc[1][1], 0, 0, 0);
                if (MODE & 2) bq[g % FD] = (g + FD < 16) ? loadb(it, g + FD) : loadb(it + 1, g + FD - 16);
            }
        }
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) for (int i = 0; i < 4; ++i) s += acc[r][c][i];
    }
    for (int g = 0; g < FD; ++g) s += bq[g].x;
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int V, int FD>
static void run(const char *name, const float *w, float *out, int grid, int wgs) {
    const int iters = 64;
    const size_t lds = 64 * 132 * 4 + (wgs == 1 ? 60000 : 0);      // (padded so that one workgroup fills a CU's LDS share when asked)
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate_kernel<MODE, V, FD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<MODE, V, FD>), dim3(grid), dim3(256), lds, 0, w, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<MODE, V, FD>), dim3(grid), dim3(256), lds, 0, w, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 16 * 4 * 4096.0;
    printf("%-58s FD %d grid %5d  %8.1f us  %6.1f TFLOP/s (%.1f%%)  %.1f cycles per 4096 FLOP at 2.4 GHz\n", name, FD, grid, ms * 1e3, flop / ms / 1e9,
           flop / ms / 1e9 / 157.3 * 100, ms * 1e-3 * 2.4e9 / (iters * 64.0 * ((grid + 255) / 256)));
}

int main() {
    order_check();
    float *w, *out;
    hipMalloc(&w, 4 << 20); hipMemset(w, 0, 4 << 20); hipMalloc(&out, 64 << 20);
    for (int wgs = 1; wgs <= 2; ++wgs) {
        printf("--- %d workgroup(s) of 4 waves per CU resident ---\n", wgs);
        const int grid = 256 * wgs;
        run<0, 0, 4>("32x32x2 one chain, registers only", w, out, grid, wgs);
        run<1, 0, 4>("32x32x2 one chain, A lds", w, out, grid, wgs);
        run<2, 0, 4>("32x32x2 one chain, B ring", w, out, grid, wgs);
        run<3, 0, 4>("32x32x2 one chain, A lds + B ring (the tile today)", w, out, grid, wgs);
        run<0, 1, 4>("16x16x4 four chains, registers only", w, out, grid, wgs);
        run<1, 1, 4>("16x16x4 four chains, A lds b64", w, out, grid, wgs);
        run<2, 1, 4>("16x16x4 four chains, B ring", w, out, grid, wgs);
        run<3, 1, 4>("16x16x4 four chains, A lds b64 + B ring", w, out, grid, wgs);
        run<3, 2, 4>("16x16x4 four chains, A lds b128 per two groups + B ring", w, out, grid, wgs);
        run<3, 2, 8>("16x16x4 four chains, A lds b128 per two groups + B ring", w, out, grid, wgs);
    }
    return 0;
}
