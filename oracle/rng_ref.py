"""Oracle RNG streams (test infrastructure, see oracle/__init__.py).

Two stream definitions exist for the grow loop.

``LegacyStream``  -- the reference's own order.  test_region_grow.py:21 seeds the
    global legacy ``numpy.random`` once and the loop consumes it in program
    order: choice (inlier) :238/:240, choice (neighbour) :250/:252,
    random(512) :266, random(512) :267.  One ``RandomState`` per room here.

``CounterStream`` -- a build extension for batched execution: a counter-based
    Philox4x32-10 stream keyed by (seed, room) and indexed by
    (seed point, restart, step, purpose, slot), so that every draw is a pure
    function of *where* it is used and not of the order instances are stepped
    in.  The HIP kernels implement bit-identical integer arithmetic
    (learn_region_grow_amd/csrc/lrg_rng.h).
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

PURPOSE_INLIER = 0
PURPOSE_NEIGHBOR = 1
PURPOSE_ADD = 2
PURPOSE_RMV = 3
PURPOSE_PERMKEY = 0x80  # or-ed with the sample purpose for the Feistel round keys


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All arguments broadcastable uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK32 for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0 = p0 >> np.uint64(32)
        lo0 = p0 & MASK32
        hi1 = p1 >> np.uint64(32)
        lo1 = p1 & MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def _fmix32(h):
    h = np.asarray(h, dtype=np.uint64) & MASK32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & MASK32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & MASK32
    h ^= h >> np.uint64(16)
    return h


def feistel_permute(j, n, keys):
    """Bijection of [0,n) evaluated at j (array), cycle-walking a 4-round
    balanced Feistel network over 2*hb bits (2^(2*hb) >= n)."""
    j = np.asarray(j, dtype=np.uint64)
    bits = max(2, int(n - 1).bit_length())
    hb = (bits + 1) // 2
    hmask = np.uint64((1 << hb) - 1)
    out = j.copy()
    todo = np.ones(out.shape, dtype=bool)
    while todo.any():
        x = out[todo]
        left = x >> np.uint64(hb)
        right = x & hmask
        for r in range(4):
            f = _fmix32(right ^ np.uint64(int(keys[r]))) & hmask
            left, right = right, left ^ f
        x = (left << np.uint64(hb)) | right
        out[todo] = x
        todo_idx = np.nonzero(todo)[0]
        todo[todo_idx[x < np.uint64(n)]] = False
    return out.astype(np.int64)


class LegacyStream:
    """Reference-order stream: one legacy RandomState per room."""
    kind = 'legacy'

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def sample(self, n, k, purpose=None, ctx=None):
        # test_region_grow.py:237-240 / :249-252
        if n >= k:
            return np.asarray(self.rs.choice(n, k, replace=False), dtype=np.int64)
        return np.asarray(list(range(n)) + list(self.rs.choice(n, k - n, replace=True)), dtype=np.int64)

    def uniform(self, k, purpose=None, ctx=None):
        # test_region_grow.py:266-267 (float64 uniforms)
        return self.rs.random_sample(k)


class CounterStream:
    """Counter-based stream; ctx = (seed_point, restart, step)."""
    kind = 'counter'

    def __init__(self, seed, room_id):
        self.k0 = int(seed) & 0xFFFFFFFF
        self.k1 = int(room_id) & 0xFFFFFFFF

    def _raw(self, k, purpose, ctx, block_base=0):
        seed_point, restart, step = ctx
        j = np.arange(k, dtype=np.uint64)
        blk = (j >> np.uint64(2)) + np.uint64(block_base)
        w = philox4x32_10(blk, np.uint64(step), np.uint64(seed_point),
                          np.uint64((purpose & 0xFF) | ((restart & 0xFFFFFF) << 8)), self.k0, self.k1)
        w = np.stack(w, axis=1)  # [k,4]
        return w[np.arange(k), (j & np.uint64(3)).astype(np.int64)].astype(np.uint64)

    def sample(self, n, k, purpose, ctx):
        if n >= k:
            keys = self._raw(4, purpose | PURPOSE_PERMKEY, ctx)
            return feistel_permute(np.arange(k), n, keys)
        x = self._raw(k, purpose, ctx)
        idx = ((x * np.uint64(n)) >> np.uint64(32)).astype(np.int64)
        idx[:n] = np.arange(n)
        return idx

    def uniform(self, k, purpose, ctx):
        x = self._raw(k, purpose, ctx)
        return ((x >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24))
