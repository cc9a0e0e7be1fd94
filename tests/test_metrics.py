"""CPU: the host metric block (learn_region_grow_amd.metrics) against the oracle restatement of
test_region_grow.py:319-355 and against the numbers the reference script itself printed (goldens)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from learn_region_grow_amd import metrics
from oracle import metrics_ref


@pytest.mark.parametrize('name', ['greedy_room100', 'greedy_room101', 'restart_room103'])
def test_metrics_match_reference_output(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    m = metrics.room_metrics(g['obj_id'], g['filled_label'])
    np.testing.assert_allclose([m['nmi'], m['ami'], m['ars'], m['prc'], m['rcl'], m['iou']], g['metrics'], rtol=1e-9)
    assert metrics.room_line(5, 0, m).startswith('Area 5 room 0 NMI: ')


def test_metrics_equal_oracle_on_random_labelings():
    rs = np.random.RandomState(0)
    for trial in range(20):
        n = rs.randint(50, 400)
        obj = rs.randint(1, rs.randint(2, 12), n)
        lab = rs.randint(1, rs.randint(2, 15), n)
        if trial % 3 == 0:
            lab = np.where(rs.rand(n) < 0.7, obj, lab)          # mostly-correct labelings exercise the matching
        a = metrics.room_metrics(obj, lab, with_sklearn=False)
        b = metrics_ref.room_metrics(obj, lab, with_sklearn=False)
        assert a['prc'] == b['prc'] and a['rcl'] == b['rcl'] and a['iou'] == b['iou']
        np.testing.assert_array_equal(a['cluster_label2'], b['cluster_label2'])


def test_aggregate_line_format():
    ms = [dict(nmi=0.8, ami=0.7, ars=0.7, prc=0.3, rcl=0.6, iou=0.5), dict(nmi=0.9, ami=0.8, ars=0.8, prc=0.2, rcl=0.6, iou=0.6)]
    assert metrics.aggregate_line(ms) == 'NMI: 0.85+-0.05 AMI: 0.75+-0.05 ARS: 0.75+-0.05 PRC 0.25+-0.05 RCL 0.60+-0.00 IOU 0.55+-0.05'
