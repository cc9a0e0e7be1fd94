// Feasibility probe (not part of the library): TWO persistent kernels of different shapes resident together and talking through memory -- what a free-running launch
// with its front workgroups (1024 threads, 128 VGPRs) and its tile teams (768 threads = three teams of ~168 VGPRs) as kernels of their own would need.  Kernel A
// (32 workgroups x 1024 threads) and kernel B (216 x 768, 164 VGPRs forced) are launched back to back on two streams; every workgroup reports in and waits (bounded:
// 20 ms) until all 248 of BOTH kernels have; then each A workgroup bounces a flag with a B workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/two_kernel tools/two_kernel_rendezvous.hip && /tmp/two_kernel
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ int ld(const int *p) { return __hip_atomic_load(const_cast<int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int THREADS, bool FAT>
__global__ __launch_bounds__(THREADS) void k(int *ctl, int all, int partner0, int iters, long long *out, int first) {
    if (FAT) asm volatile("" ::: "v163");      // (account 164 VGPRs: three 768-thread waves per SIMD at most)
    const long long t0 = wall_clock64();
    int ok = 1;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ld(&ctl[0]) < all) { if (wall_clock64() - t0 > 2000000) { ok = 0; break; } __builtin_amdgcn_s_sleep(8); }
        if (!ok) st(&ctl[16], 1);
        if (blockIdx.x == 0) out[first ? 0 : 1] = wall_clock64() - t0;      // how long workgroup 0 of this kernel waited for everybody
        if (ok && first && (int)blockIdx.x < partner0) {          // kernel A's workgroup b bounces a flag with kernel B's workgroup b
            int *f = ctl + 64 + 32 * blockIdx.x;
            const long long t1 = wall_clock64();
            for (int i = 0; i < iters; ++i) { st(&f[0], 2 * i + 1); while (ld(&f[16]) != 2 * i + 2) { if (wall_clock64() - t1 > 2000000) { i = iters; break; } } }
            if (blockIdx.x == 0) out[2] = wall_clock64() - t1;
        } else if (ok && !first && (int)blockIdx.x < partner0) {
            int *f = ctl + 64 + 32 * blockIdx.x;
            const long long t1 = wall_clock64();
            for (int i = 0; i < iters; ++i) { while (ld(&f[0]) != 2 * i + 1) { if (wall_clock64() - t1 > 2000000) { i = iters; break; } } st(&f[16], 2 * i + 2); }
        }
    }
    __syncthreads();
}

int main() {
    int *ctl; long long *out;
    hipMalloc(&ctl, 1 << 20); hipMalloc(&out, 64);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(ctl, 0, 1 << 20); hipMemset(out, 0, 64); hipDeviceSynchronize();
        const int na = 32, nb = 216, iters = 1000;      // (workgroups go round the 8 XCDs in turn: 4 + 27 per XCD, one per CU)
        hipLaunchKernelGGL((k<1024, false>), dim3(na), dim3(1024), 60000, sa, ctl, na + nb, na, iters, out, 1);
        hipLaunchKernelGGL((k<768, true>), dim3(nb), dim3(768), 100000, sb, ctl, na + nb, na, iters, out, 0);
        hipDeviceSynchronize();
        long long h[3]; int hc[17];
        hipMemcpy(h, out, 24, hipMemcpyDeviceToHost); hipMemcpy(hc, ctl, sizeof(hc), hipMemcpyDeviceToHost);
        printf("run %d: %d of %d workgroups reported in; kernel A's workgroup 0 waited %.1f us, kernel B's %.1f us for everybody%s; a flag round trip between the kernels %.2f us\n",
               rep, hc[0], na + nb, h[0] / 100.0, h[1] / 100.0, hc[16] ? " (TIMED OUT: the kernels were not resident together)" : "", h[2] / 100.0 / iters);
    }
    return 0;
}
