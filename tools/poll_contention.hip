// Microbenchmark (not part of the library): what idle pollers cost a hand-over.  Two workgroups bounce a flag (device-scope loads / stores, as tools/xcd_pingpong.hip)
// while `pollers` wavefronts (one lane each, s_sleep(8) between looks: the free-running kernel's idle tile teams) poll words `stride` bytes apart that nobody writes
// -- next to the flags' lines (the ring entries of consecutive tickets: 4 bytes apart) or spread over the memory channels.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/poll_contention tools/poll_contention.hip && /tmp/poll_contention
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ int ld(const int *p) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(p), 0, 0x7fffffff, 0x00020000);
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 16);
}
__device__ __forceinline__ void st(int *p, int v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)v, r, 0, 0, 16);
}

__global__ void k(int *flags, int *poll, long stride_words, int pollers, int iters, long long *out, int sleep) {
    const int wave = (int)(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64);
    if ((threadIdx.x & 63) != 0) return;
    const long long t0 = wall_clock64();
    if (wave == 0) {
        for (int i = 0; i < iters; ++i) {
            st(&flags[0], 2 * i + 1);
            while (ld(&flags[16]) != 2 * i + 2) { if (wall_clock64() - t0 > 20000000) { out[1] = i + 1; break; } }
            if (out[1]) break;
        }
        out[0] = wall_clock64() - t0;
        st(&flags[32], 1);
    } else if (wave == 4) {      // (another workgroup)
        for (int i = 0; i < iters; ++i) {
            while (ld(&flags[0]) != 2 * i + 1) { if (wall_clock64() - t0 > 20000000) return; }
            st(&flags[16], 2 * i + 2);
        }
    } else if (wave >= 8 && wave < 8 + pollers) {
        const int *w = poll + (long)(wave - 8) * stride_words;
        long long n = 0;
        while (ld(w) == 0 && ld(&flags[32]) == 0 && wall_clock64() - t0 < 30000000) { ++n; if (sleep) __builtin_amdgcn_s_sleep(8); }
        if (wave == 8) out[2] = n;
    }
}

int main() {
    int *buf; long long *out;
    const size_t bytes = 64u << 20;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    const int iters = 2000;
    struct { const char *name; int pollers; long stride; int sleep; } cases[] = {
        {"no pollers", 0, 1, 1},
        {"64 pollers, words 4 B apart, next to the flags", 64, 1, 1},
        {"750 pollers, words 4 B apart, next to the flags", 750, 1, 1},
        {"750 pollers, words 4 B apart, 32 MB away from the flags", 750, 1, 1},
        {"750 pollers, words 272 B apart", 750, 68, 1},
        {"750 pollers, words 4352 B apart", 750, 1088, 1},
        {"750 pollers, 4 B apart, no sleep between looks", 750, 1, 0},
        {"68 pollers, 64 B apart, no sleep (the fronts' own polls)", 68, 16, 0},
    };
    int ci = 0;
    for (auto &c : cases) {
        hipMemset(buf, 0, bytes); hipMemset(out, 0, 64);
        int *flags = buf;
        int *poll = (ci == 3) ? buf + (32u << 20) / 4 : buf + 64;       // (right behind the flags' lines, or far away)
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, flags, poll, c.stride, c.pollers, iters, out, c.sleep);
        hipDeviceSynchronize();
        long long h[3];
        hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
        if (h[1]) printf("%-64s TIMED OUT in round %lld\n", c.name, h[1]);
        else printf("%-64s %.2f us per round trip (two hand-overs); a poller looked %lld times\n", c.name, (double)h[0] / 100.0 / iters, h[2]);
        ++ci;
    }
    return 0;
}
