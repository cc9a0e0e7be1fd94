"""Golden outputs of the reference's own savePLY / savePCD (learn_region_grow_util.py:33-73), produced by importing the
unmodified reference module in this container through the TensorFlow stand-in of make_golden.py:
    python tests/golden/make_format_golden.py
Also copies two small DATA files of the reference's model directory that pin the checkpoint reader: the index of the real
LrgNet checkpoint (names / shapes / offsets / CRCs of its 99 variables) and, from the MCPNet checkpoint whose data blob is
present, the index plus the raw bytes of every variable of at most 4 KiB.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import tf_numpy_standin  # noqa: E402

tf_numpy_standin.install()
sys.path.insert(0, REF)
import learn_region_grow_util as ref  # noqa: E402

rs = np.random.RandomState(3)
pts = np.zeros((40, 6))
pts[:, :3] = rs.randn(40, 3) * 3.3
pts[:, 3:6] = rs.randint(0, 256, (40, 3))
pts[0, :3] = [0, -0.0000004, 123456.789]
with contextlib.redirect_stdout(io.StringIO()):
    ref.savePLY(os.path.join(HERE, 'ref_savePLY.ply'), pts)
    ref.savePCD(os.path.join(HERE, 'ref_savePCD.pcd'), pts)
np.save(os.path.join(HERE, 'ref_save_points.npy'), pts)

from learn_region_grow_amd import checkpoint as ck  # noqa: E402

idx = open(os.path.join(REF, 'models/lrgnet_model5.ckpt.index'), 'rb').read()
open(os.path.join(HERE, 'lrgnet_model5.ckpt.index'), 'wb').write(idx)
prefix = os.path.join(REF, 'models/mcpnet_model5.ckpt')
_, entries = ck.read_bundle_index(prefix)
data = open(prefix + '.data-00000-of-00001', 'rb').read()
small = {n: e for n, e in entries.items() if e.size <= 4096}
np.savez(os.path.join(HERE, 'mcpnet_bundle_small.npz'),
         index=np.frombuffer(open(prefix + '.index', 'rb').read(), dtype=np.uint8),
         names=np.array(sorted(small)),
         **{'raw_%d' % i: np.frombuffer(data[small[n].offset:small[n].offset + small[n].size], dtype=np.uint8)
            for i, n in enumerate(sorted(small))})
print('wrote', sorted(small))
