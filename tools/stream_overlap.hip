// Do kernels on different HIP streams overlap on this GPU?  K dependent launches of a G-workgroup kernel that spins T us,
// on S streams at once: wall time ~ K * T if the streams run side by side, ~ S * K * T if they take turns.
//   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/stream_overlap tools/stream_overlap.hip && /tmp/stream_overlap 0   (1: CU-masked streams)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(long long ticks, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *sink = 1;
}

static double run(int S, int K, int G, int us, bool masked) {
    std::vector<hipStream_t> st(S);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    for (int s = 0; s < S; ++s) {
        if (masked) {
            std::vector<uint32_t> m(words, 0);
            for (int b = s * ncu / S; b < (s + 1) * ncu / S; ++b) m[b / 32] |= 1u << (b % 32);
            if (hipExtStreamCreateWithCUMask(&st[s], words, m.data()) != hipSuccess) { printf("mask stream failed\n"); return -1; }
        } else hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
    }
    for (int s = 0; s < S; ++s) hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, st[s], 100, nullptr);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < K; ++k)
        for (int s = 0; s < S; ++s) hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, st[s], (long long)us * 100, nullptr);
    hipDeviceSynchronize();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int s = 0; s < S; ++s) hipStreamDestroy(st[s]);
    return dt;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int K = 400;
    const int m0 = argc > 1 ? atoi(argv[1]) : 0;
    const int smax = argc > 2 ? atoi(argv[2]) : 4;        // streams 1 .. smax
    for (int masked = m0; masked <= m0; ++masked)
        for (int G : {64, 248})
            for (int us : {5, 20})
                for (int S = 1; S <= smax; ++S) {
                    const double dt = run(S, K, G, us, masked);
                    printf("%s streams %d  grid %3d  spin %2d us: %7.1f us per round of %d launches (%.2f x one stream's kernel time)\n",
                           masked ? "CU-masked" : "plain    ", S, G, us, dt / K * 1e6, S, dt / K * 1e6 / us);
                }
    return 0;
}
