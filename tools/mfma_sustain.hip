// Microbenchmark (not part of the library): what the fp32 matrix cores SUSTAIN -- v_mfma_f32_32x32x2_f32 from registers only, two workgroups of four wavefronts per CU, in
// launches of ~0.4 / 3 / 25 ms back to back, with the shader clock read inside the kernel (s_memtime cycles against the 100 MHz wall clock).  tools/mfma_peak.hip's
// 99 % of 157.3 TFLOP/s is a 0.4 ms launch on an idle chip.      hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_sustain tools/mfma_sustain.hip && /tmp/mfma_sustain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool RANDOM>
__global__ __launch_bounds__(256, 2) void k(float *out, long long *clk, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // RANDOM: operands with full-entropy mantissas, different in every lane and k-step (what real activations and weights toggle in the multipliers);
    // else one small constant per lane
    float av[16], bv[16];
    unsigned h = (threadIdx.x + 1u) * 2654435761u + blockIdx.x * 40503u;
    for (int g = 0; g < 16; ++g) {
        h = h * 1664525u + 1013904223u; av[g] = RANDOM ? __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f : (float)(threadIdx.x & 7) * 0.125f;
        h = h * 1664525u + 1013904223u; bv[g] = RANDOM ? __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f : 0.5f;
    }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g], bv[(g + t) & 15], acc[t], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
    float *out; long long *clk, h[2];
    hipMalloc(&out, 64 << 20); hipMalloc(&clk, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep)
        for (int iters : {256, 2048, 16384, 16384, 256}) {
            hipEventRecord(e0);
            if (rep & 1) hipLaunchKernelGGL(k<true>, dim3(512), dim3(256), 0, 0, out, clk, iters);
            else hipLaunchKernelGGL(k<false>, dim3(512), dim3(256), 0, 0, out, clk, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double flop = 512.0 * 4 * iters * 64 * 4096.0;
            printf("%s iters %6d  %9.1f us  %7.1f TFLOP/s (%.1f %% of 157.3)   shader clock %.0f MHz (s_memtime cycles / 100 MHz wall clock), MFMA issue %.1f %% of the wave's cycles x 2 waves\n",
                   (rep & 1) ? "random operands  " : "constant operands", iters, ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100, (double)h[0] / ((double)h[1] / 100.0), 100.0 * iters * 64 * 64 / (double)h[0] * 2);
        }
    return 0;
}
