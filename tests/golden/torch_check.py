"""An independent evaluation of the LrgNet graph with torch.nn.functional.conv1d on the CPU, against golden activations made by
the reference's own LrgNet.__init__ under the NumPy stand-in (where tf.nn.conv1d is ``x @ W[0]`` by definition).  Used by
make_golden.py when a golden is written and by tests/test_oracle_golden.py."""
import numpy as np


def check_against_torch(g, w):
    """g: mapping with the golden's arrays, w: name -> weight arrays ([1,Cin,Cout] kernels).  Returns the largest error relative to
    each tensor's own scale; raises on a shape / layout mismatch."""
    import torch
    import torch.nn.functional as TF

    def layer(x, pre, i):        # tf.nn.conv1d(stride 1, VALID, filter [1,Cin,Cout]) + bias_add, through torch's NCW conv1d
        k = torch.from_numpy(np.ascontiguousarray(w['%skernel%d' % (pre, i)][0].T[:, :, None]))          # [Cout, Cin, 1]
        return (TF.conv1d(x.transpose(1, 2), k) + torch.from_numpy(np.asarray(w['%sbias%d' % (pre, i)]))[None, :, None]).transpose(1, 2)
    keys = list(g.keys()) if hasattr(g, 'keys') else list(g.files)
    nconv = sum(1 for k in keys if k.startswith('conv'))
    nhead = sum(1 for k in keys if k.startswith('add_conv'))
    worst = 0.0

    def cmp(got, want):
        nonlocal worst
        want = np.asarray(want)
        assert tuple(got.shape) == tuple(want.shape), (got.shape, want.shape)
        worst = max(worst, float(np.abs(got.numpy() - want).max()) / max(1.0, float(np.abs(want).max())))
    acts = {}
    for pre, key, src in (('lrg_', 'conv', 'inlier'), ('lrg_neighbor_', 'neighbor_conv', 'neighbor')):
        h = torch.from_numpy(np.asarray(g[src]))
        for i in range(nconv):
            h = torch.relu(layer(h, pre, i))
            cmp(h, g['%s%d' % (key, i)])
            acts['%s%d' % (key, i)] = h
    pooled = torch.cat([acts['conv%d' % (nconv - 1)].max(dim=1).values, acts['neighbor_conv%d' % (nconv - 1)].max(dim=1).values], dim=1)
    cmp(pooled, g['pooled'])
    for pre, key, loc, out in (('lrg_add_', 'add_conv', 'neighbor_conv1', 'add_output'), ('lrg_remove_', 'remove_conv', 'conv1', 'remove_output')):
        local = acts[loc]
        h = torch.cat([pooled[:, None, :].expand(-1, local.shape[1], -1), local], dim=2)          # tile + concat, pooled first (:128-135)
        for i in range(nhead):
            h = torch.relu(layer(h, pre, i))
            cmp(h, g['%s%d' % (key, i)])
        cmp(layer(h, pre, nhead), g[out])
    return worst
