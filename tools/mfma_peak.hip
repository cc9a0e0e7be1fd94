// Microbenchmark (not part of the library): sustained fp32 MFMA rate of v_mfma_f32_32x32x2_f32 on this device with the
// operand streams of the fused LrgNet kernel added one at a time.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE bit0: A from LDS (ds_read_b128 per tile per group); bit1: B from global through a ring of FD k-groups that never
// drains (refilled across iterations); bit2: B pre-packed so that a lane's four values are one dwordx4 load.
template <int MODE, int RT, int FD>
__global__ __launch_bounds__(256, 2) void k(const float *w, float *out, int iters, int ldw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 64 * 132; i += 256) smem[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[RT], acc2;
    for (int t = 0; t < RT; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
    const float *ap = smem + li * 132 + 4 * lh;
    float4 a[RT];
    for (int t = 0; t < RT; ++t) a[t] = make_float4(1.f, 2.f, 3.f, 4.f);
    float4 bq[FD];
    const int loff = 4 * lh * ldw + li;
    auto loadb = [&](const float *wrow, int g) -> float4 {
        if (MODE & 4) return *reinterpret_cast<const float4 *>(wrow + (long)g * 256 + lane * 4);     // packed: [group][lane][4]
        const float *w0 = wrow + (long)(8 * g) * ldw;
        return make_float4(w0[loff], w0[loff + ldw], w0[loff + 2 * ldw], w0[loff + 3 * ldw]);
    };
    auto wrow_of = [&](int it) { return (MODE & 4) ? w + ((it & 3) * 4 + wn) * 16 * 256 : w + wn * 32 + (it & 3) * 128; };
    for (int g = 0; g < FD; ++g) bq[g] = (MODE & 2) ? loadb(wrow_of(0), g) : make_float4(1.f, 2.f, 0.5f, 0.25f);
    f32x4 c4[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 4; ++i) c4[t][i] = 0.f;
    const float *ap16 = smem + (lane & 15) * 132 + 2 * (lane >> 4);
    float2 a2[3][2];
    for (int u = 0; u < 3; ++u) { a2[u][0] = make_float2(1.f, 2.f); a2[u][1] = make_float2(3.f, 4.f); }
    for (int it = 0; it < iters; ++it) {
        const float *wrow = wrow_of(it), *wnext = wrow_of(it + 1);
        if ((MODE & 16) && (MODE & 1)) {
            a2[0][0] = *reinterpret_cast<const float2 *>(ap16); a2[0][1] = *reinterpret_cast<const float2 *>(ap16 + 16 * 132);
            a2[1][0] = *reinterpret_cast<const float2 *>(ap16 + 8); a2[1][1] = *reinterpret_cast<const float2 *>(ap16 + 16 * 132 + 8);
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if ((MODE & 1) && !(MODE & 16))
#pragma unroll
                for (int t = 0; t < RT; ++t) a[t] = *reinterpret_cast<const float4 *>(ap + t * 32 * 132 + 8 * g);
            const float4 b = bq[g % FD];
            if (MODE & 16) {      // v_mfma_f32_16x16x4_f32: the wave's 32 x 32 strip as 2 x 2 blocks = four accumulator chains; A as one ds_read_b64 per row half
                                  // and k-group (two groups ahead), B the same dwordx4 per lane and k-group
                if (MODE & 1) {
                    if (g + 2 < 16) { a2[(g + 2) % 3][0] = *reinterpret_cast<const float2 *>(ap16 + 8 * (g + 2)); a2[(g + 2) % 3][1] = *reinterpret_cast<const float2 *>(ap16 + 16 * 132 + 8 * (g + 2)); }
                }
                const float2 x0 = a2[g % 3][0], x1 = a2[g % 3][1];
                c4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, b.x, c4[0], 0, 0, 0);
                c4[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, b.z, c4[1], 0, 0, 0);
                c4[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, b.x, c4[2], 0, 0, 0);
                c4[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, b.z, c4[3], 0, 0, 0);
                c4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, b.y, c4[0], 0, 0, 0);
                c4[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, b.w, c4[1], 0, 0, 0);
                c4[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, b.y, c4[2], 0, 0, 0);
                c4[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, b.w, c4[3], 0, 0, 0);
                if (MODE & 2) bq[g % FD] = (g + FD < 16) ? loadb(wrow, g + FD) : loadb(wnext, g + FD - 16);
                continue;
            }
            if (MODE & 8) {       // one tile, two accumulator chains: the MFMAs of a k-group alternate between them
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].x, b.x, acc[0], 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].y, b.y, acc2, 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].z, b.z, acc[0], 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].w, b.w, acc2, 0, 0, 0);
                if (MODE & 2) bq[g % FD] = (g + FD < 16) ? loadb(wrow, g + FD) : loadb(wnext, g + FD - 16);
                continue;
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b.x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b.y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b.z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b.w, acc[t], 0, 0, 0);
            if (MODE & 2) bq[g % FD] = (g + FD < 16) ? loadb(wrow, g + FD) : loadb(wnext, g + FD - 16);
        }
        constexpr int DA = (MODE & 1) ? 2 : 0;
        constexpr int NV = (MODE & 2) ? ((MODE & 4) ? 1 : 4) : 0;
        if (MODE & 16) {
            if (DA) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (DA && g + 2 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                if (NV) __builtin_amdgcn_sched_group_barrier(0x020, NV, 0);
            }
            continue;
        }
        if (DA) __builtin_amdgcn_sched_group_barrier(0x100, RT * DA, 0);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0);
            if (NV) __builtin_amdgcn_sched_group_barrier(0x020, NV, 0);
            if (DA && g + DA < 16) __builtin_amdgcn_sched_group_barrier(0x100, RT, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < RT; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i] + acc2[i];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 4; ++i) s += c4[t][i];
    for (int g = 0; g < FD; ++g) s += bq[g].x;
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int RT, int FD>
static void run(const char *name, const float *w, float *out, int grid, int wgs_note) {
    const int iters = 64;
    const size_t lds = 64 * 132 * 4 + (wgs_note == 1 ? 60000 : 0);   // pad LDS to force 1 WG/CU when asked
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, RT, FD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, RT, FD>), dim3(grid), dim3(256), lds, 0, w, out, iters, 512);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, RT, FD>), dim3(grid), dim3(256), lds, 0, w, out, iters, 512);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * 16 * 4 * RT * 4096.0;
    printf("%-44s FD %d grid %6d  %8.1f us  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", name, FD, grid, ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

int main() {
    float *w, *out;
    hipMalloc(&w, 4 << 20); hipMemset(w, 0, 4 << 20); hipMalloc(&out, 64 << 20);
    for (int wgs = 1; wgs <= 2; ++wgs) {
        printf("--- %d workgroup(s) of 4 waves per CU resident ---\n", wgs);
        for (int grid : {256 * wgs, 256 * wgs * 8}) {
            run<0, 2, 4>("RT=2 registers only", w, out, grid, wgs);
            run<1, 2, 4>("RT=2 A lds", w, out, grid, wgs);
            run<3, 2, 4>("RT=2 A lds + B 4x dword ring", w, out, grid, wgs);
            run<3, 2, 8>("RT=2 A lds + B 4x dword ring", w, out, grid, wgs);
            run<7, 2, 4>("RT=2 A lds + B packed dwordx4 ring", w, out, grid, wgs);
            run<7, 2, 8>("RT=2 A lds + B packed dwordx4 ring", w, out, grid, wgs);
            run<3, 1, 4>("RT=1 A lds + B 4x dword ring", w, out, grid, wgs);
            run<3, 1, 8>("RT=1 A lds + B 4x dword ring", w, out, grid, wgs);
            run<7, 1, 4>("RT=1 A lds + B packed dwordx4 ring", w, out, grid, wgs);
            run<7, 1, 8>("RT=1 A lds + B packed dwordx4 ring", w, out, grid, wgs);
            run<0, 1, 4>("RT=1 registers only, one chain", w, out, grid, wgs);
            run<8, 1, 4>("RT=1 registers only, two chains", w, out, grid, wgs);
            run<15, 1, 4>("RT=1 A lds + B packed ring, two chains", w, out, grid, wgs);
            run<16, 1, 4>("16x16x4 2x2 blocks, registers only", w, out, grid, wgs);
            run<17, 1, 4>("16x16x4 2x2 blocks, A lds b64", w, out, grid, wgs);
            run<23, 1, 4>("16x16x4 2x2 blocks, A lds b64 + B packed ring", w, out, grid, wgs);
            run<23, 1, 8>("16x16x4 2x2 blocks, A lds b64 + B packed ring", w, out, grid, wgs);
        }
    }
    return 0;
}
