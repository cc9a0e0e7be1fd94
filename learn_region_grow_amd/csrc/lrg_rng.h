// Counter-based RNG for batched region growing (device side).
// Bit-identical integer arithmetic to oracle/rng_ref.py (CounterStream): Philox4x32-10 keyed by
// (rng_seed, room_id), counter = (slot/4, step, seed_point, purpose | restart<<8); subset sampling
// without replacement through a cycle-walking 4-round Feistel permutation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LRG_PURPOSE_INLIER 0u
#define LRG_PURPOSE_NEIGHBOR 1u
#define LRG_PURPOSE_ADD 2u
#define LRG_PURPOSE_RMV 3u
#define LRG_PURPOSE_PERMKEY 0x80u

struct lrg_u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ lrg_u32x4 lrg_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                       uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    lrg_u32x4 o = {c0, c1, c2, c3};
    return o;
}

// Word for draw `j` of a purpose at (seed_point, restart, step).
__device__ __forceinline__ uint32_t lrg_rng_word(uint32_t j, uint32_t purpose, uint32_t seed_point, uint32_t restart,
                                                 uint32_t step, uint32_t k0, uint32_t k1) {
    lrg_u32x4 r = lrg_philox4x32_10(j >> 2, step, seed_point, (purpose & 0xFFu) | ((restart & 0xFFFFFFu) << 8), k0, k1);
    uint32_t l = j & 3u;
    return l == 0 ? r.x : (l == 1 ? r.y : (l == 2 ? r.z : r.w));
}

__device__ __forceinline__ uint32_t lrg_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// Bijection of [0,n) at j (n >= 2): balanced Feistel over 2*hb bits, cycle-walking.
__device__ __forceinline__ uint32_t lrg_feistel_permute(uint32_t j, uint32_t n, lrg_u32x4 keys) {
    uint32_t bits = 32u - (uint32_t)__clz((int)(n - 1u));
    if (bits < 2u) bits = 2u;
    uint32_t hb = (bits + 1u) >> 1;
    uint32_t hmask = (1u << hb) - 1u;
    uint32_t x = j;
    do {
        uint32_t left = x >> hb, right = x & hmask, t;
        t = left ^ (lrg_fmix32(right ^ keys.x) & hmask); left = right; right = t;
        t = left ^ (lrg_fmix32(right ^ keys.y) & hmask); left = right; right = t;
        t = left ^ (lrg_fmix32(right ^ keys.z) & hmask); left = right; right = t;
        t = left ^ (lrg_fmix32(right ^ keys.w) & hmask); left = right; right = t;
        x = (left << hb) | right;
    } while (x >= n);
    return x;
}

// Position (into a compacted list of n entries) of sample slot j out of k   (test_region_grow.py:237-240)
__device__ __forceinline__ uint32_t lrg_sample_position(uint32_t j, uint32_t n, uint32_t k, uint32_t purpose,
                                                        uint32_t seed_point, uint32_t restart, uint32_t step,
                                                        uint32_t k0, uint32_t k1) {
    if (n >= k) {
        lrg_u32x4 keys = lrg_philox4x32_10(0u, step, seed_point,
                                           ((purpose | LRG_PURPOSE_PERMKEY) & 0xFFu) | ((restart & 0xFFFFFFu) << 8), k0, k1);
        return lrg_feistel_permute(j, n, keys);
    }
    if (j < n) return j;
    uint32_t w = lrg_rng_word(j, purpose, seed_point, restart, step, k0, k1);
    return (uint32_t)(((uint64_t)w * (uint64_t)n) >> 32);
}

__device__ __forceinline__ float lrg_uniform01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }
