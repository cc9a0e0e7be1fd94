// Microbenchmark (not part of the library): the round trip of a flag between two workgroups -- on the same XCD through its L2 (sc0 loads: the L1 bypassed, plain
// write-through stores) against the device-scope hand-over the free-running kernel uses (sc1: through memory), same and different XCDs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_pingpong tools/xcd_pingpong.hip && /tmp/xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ __forceinline__ int ld(const int *p) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(p), 0, 0x7fffffff, 0x00020000);
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void st(int *p, int v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)v, r, 0, 0, AUX);
}

// workgroup `a` and workgroup `b` bounce a counter: a writes 2i+1 to flag[0], b answers 2i+2 in flag[16]
template <int LAUX, int SAUX>
__global__ void k(int *flags, int a, int b, int iters, long long *out, int *xcc) {
    int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[blockIdx.x] = id & 0xf;
    if (threadIdx.x != 0) return;
    if ((int)blockIdx.x == a) {
        const long long t0 = wall_clock64();
        for (int i = 0; i < iters; ++i) {
            st<SAUX>(&flags[0], 2 * i + 1);
            while (ld<LAUX>(&flags[16]) != 2 * i + 2) { if (wall_clock64() - t0 > 5000000) { out[1] = i + 1; return; } }
        }
        out[0] = wall_clock64() - t0;
    } else if ((int)blockIdx.x == b) {
        const long long t0 = wall_clock64();
        for (int i = 0; i < iters; ++i) {
            while (ld<LAUX>(&flags[0]) != 2 * i + 1) { if (wall_clock64() - t0 > 5000000) return; }
            st<SAUX>(&flags[16], 2 * i + 2);
        }
    }
}

template <int LAUX, int SAUX>
static void run(const char *name, int a, int b) {
    int *flags, *xcc; long long *out;
    hipMalloc(&flags, 4096); hipMalloc(&xcc, 4096); hipMalloc(&out, 64);
    hipMemset(flags, 0, 4096); hipMemset(out, 0, 64);
    const int iters = 2000;
    hipLaunchKernelGGL((k<LAUX, SAUX>), dim3(32), dim3(64), 0, 0, flags, a, b, iters, out, xcc);
    hipDeviceSynchronize();
    long long h, hh[2]; int hx[32];
    hipMemcpy(hh, out, 16, hipMemcpyDeviceToHost); h = hh[0];
    if (hh[1]) { printf("%-64s TIMED OUT in round %lld (50 ms): the flag was never seen\n", name, hh[1]); return; } hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
    printf("%-64s workgroups %2d (XCC %d) <-> %2d (XCC %d): %.2f us per round trip (two hand-overs)\n", name, a, hx[a], b, hx[b], (double)h / 100.0 / iters);
    hipFree(flags); hipFree(xcc); hipFree(out);
}

int main() {
    // aux bits: 1 = sc0, 16 = sc1
    run<16, 16>("sc1 loads, sc1 stores (device scope: lrg_ld_coh / lrg_st_coh)", 0, 8);
    run<16, 16>("sc1 loads, sc1 stores (device scope)", 0, 1);
    run<17, 17>("sc0 sc1 loads and stores (system scope)", 0, 1);
    run<1, 0>("sc0 loads, plain stores (the XCD's L2)", 0, 8);
    run<1, 1>("sc0 loads, sc0 stores", 0, 8);
    run<1, 0>("sc0 loads, plain stores -- DIFFERENT XCDs (not coherent: expected to time out)", 0, 1);
    return 0;
}
