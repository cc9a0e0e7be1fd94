// Shared helpers for the liblrg_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <math.h>
#include "../../include/lrg_hip.h"

#define LRG_HIP_CHECK(expr)                                 \
    do {                                                    \
        hipError_t _e = (expr);                             \
        if (_e != hipSuccess) return -(int)_e;              \
    } while (0)

#define LRG_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t _e = hipGetLastError();                  \
        if (_e != hipSuccess) return -(int)_e;              \
    } while (0)

static inline size_t lrg_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Function attributes (dynamic-LDS cap) are per device: remember per device ordinal which kernels were raised.  A race
// between host threads only repeats an idempotent call.
#define LRG_MAX_DEVICES 64
static inline int lrg_current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= LRG_MAX_DEVICES) d = 0;
    return d;
}

__device__ __forceinline__ int lrg_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ float lrg_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ uint64_t lrg_fmix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

#define LRG_VOX_OFF (1 << 20)
#define LRG_HASH_EMPTY 0xFFFFFFFFFFFFFFFFULL

// Packed voxel key; returns LRG_HASH_EMPTY when a coordinate is outside the 21-bit window.
__device__ __forceinline__ uint64_t lrg_pack_voxel(int x, int y, int z) {
    unsigned ux = (unsigned)(x + LRG_VOX_OFF), uy = (unsigned)(y + LRG_VOX_OFF), uz = (unsigned)(z + LRG_VOX_OFF);
    if ((ux | uy | uz) >> 21) return LRG_HASH_EMPTY;
    return ((uint64_t)ux << 42) | ((uint64_t)uy << 21) | (uint64_t)uz;
}

__device__ __forceinline__ int lrg_hash_lookup(const uint64_t *keys, const int32_t *vals, int mask, uint64_t key) {
    if (key == LRG_HASH_EMPTY) return -1;
    unsigned h = (unsigned)lrg_fmix64(key) & (unsigned)mask;
    for (int probe = 0; probe <= mask; ++probe) {
        const uint64_t k = keys[h];
        const int v = vals[h];               // requested together with the key: one memory round trip per probe, not two
        if (k == key) return v;
        if (k == LRG_HASH_EMPTY) return -1;
        h = (h + 1) & (unsigned)mask;
    }
    return -1;
}

// rint(x / res) exactly as numpy.round(float32 / float32) (test_region_grow.py:175): IEEE divide, half-to-even.
__device__ __forceinline__ int lrg_voxel_of(float x, float res) { return (int)rintf(__fdiv_rn(x, res)); }

// Wavefront reductions on the DPP cross-lane network (VALU speed; __shfl_* go through the LDS crossbar at ~100 cycles a hop
// and ballot+popcount pays a VALU->SALU round trip per call).  The result is returned to every lane.
#define LRG_DPP_REDUCE(OP)                                                                                   \
    v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0xB1, 0xF, 0xF, false));  /* quad_perm [1,0,3,2] */        \
    v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x4E, 0xF, 0xF, false));  /* quad_perm [2,3,0,1] */        \
    v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x141, 0xF, 0xF, false)); /* row_half_mirror    */        \
    v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x140, 0xF, 0xF, false)); /* row_mirror         */        \
    v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xA, 0xF, false)); /* row_bcast15 -> rows 1,3 */    \
    v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xC, 0xF, false)); /* row_bcast31 -> rows 2,3 */    \
    return __builtin_amdgcn_readlane(v, 63);

__device__ __forceinline__ int lrg_op_add(int a, int b) { return a + b; }
__device__ __forceinline__ int lrg_op_umin(int a, int b) { return (int)min((unsigned)a, (unsigned)b); }
__device__ __forceinline__ int lrg_op_umax(int a, int b) { return (int)max((unsigned)a, (unsigned)b); }
__device__ __forceinline__ int lrg_wave_sum_i32(int v) { const int ident = 0; LRG_DPP_REDUCE(lrg_op_add) }
__device__ __forceinline__ unsigned lrg_wave_min_u32(unsigned x) { int v = (int)x; const int ident = -1; LRG_DPP_REDUCE(lrg_op_umin) }
__device__ __forceinline__ unsigned lrg_wave_max_u32(unsigned x) { int v = (int)x; const int ident = 0; LRG_DPP_REDUCE(lrg_op_umax) }

// Inclusive prefix sum over the 64 lanes of a wavefront on the DPP network: Hillis-Steele inside each row of 16 lanes
// (row_shr 1, 2, 4, 8; lanes without a source add 0), then the row totals ripple through row_bcast15 / row_bcast31.
// Needs every lane active.
__device__ __forceinline__ int lrg_wave_incl_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  /* row_shr:1 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  /* row_shr:2 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  /* row_shr:4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  /* row_shr:8 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  /* row_bcast15 -> rows 1, 3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  /* row_bcast31 -> rows 2, 3 */
    return v;
}
