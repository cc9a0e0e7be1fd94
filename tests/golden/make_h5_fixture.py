#!/opt/conda/bin/python3.9
"""Generate the HDF5 fixtures of tests/test_formats.py with h5py (present only in this container's conda interpreter):
    /opt/conda/bin/python3.9 tests/golden/make_h5_fixture.py
rooms_gzip.h5     -- the layout tools/generate_synthetic_rooms.py:111-115 of the reference writes: 'points' float32 [N,8] and
                     'count_room' int32 [R], both chunked + gzip level 4
rooms_variants.h5 -- the same arrays contiguous, gzip+shuffle, gzip+fletcher32, multi-chunk, plus int64 / float64 / uint8
rooms_expected.npz -- the arrays themselves
"""
import os
import h5py
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
rs = np.random.RandomState(7)
count = np.array([37, 5, 120], dtype=np.int32)
pts = np.zeros((count.sum(), 8), dtype=np.float32)
pts[:, :3] = rs.rand(len(pts), 3) * 5
pts[:, 3:6] = rs.rand(len(pts), 3) - 0.5
pts[:, 6] = rs.randint(0, 9, len(pts))
pts[:, 7] = rs.randint(0, 13, len(pts))
big = rs.randn(700, 8).astype(np.float32)
with h5py.File(os.path.join(here, 'rooms_gzip.h5'), 'w') as f:
    f.create_dataset('points', data=pts, compression='gzip', compression_opts=4, dtype=np.float32)
    f.create_dataset('count_room', data=count, compression='gzip', compression_opts=4, dtype=np.int32)
with h5py.File(os.path.join(here, 'rooms_variants.h5'), 'w') as f:
    f.create_dataset('contiguous', data=pts)
    f.create_dataset('shuffled', data=pts, compression='gzip', shuffle=True)
    f.create_dataset('checksummed', data=pts, compression='gzip', fletcher32=True)
    f.create_dataset('multichunk', data=big, chunks=(64, 3), compression='gzip', compression_opts=1)
    f.create_dataset('i64', data=np.arange(-5, 20, dtype=np.int64))
    f.create_dataset('f64', data=big[:9, :2].astype(np.float64))
    f.create_dataset('u8', data=(big[:50, 0] * 40).astype(np.uint8), chunks=(16,), compression='gzip')
    f.create_dataset('scalar', data=np.float32(2.5))
np.savez(os.path.join(here, 'rooms_expected.npz'), points=pts, count_room=count, big=big)
print('wrote fixtures')
