"""Per-room evaluation metrics of the reference (host side, NumPy + sklearn).

What /root/reference/test_region_grow.py:319-355 computes after a room is labeled: greedy IoU > 0.5 matching of
ground-truth instances (largest first) to predicted clusters, PRC / RCL / mean best IoU, and sklearn's NMI / AMI / ARS;
plus the aggregate line of :379-381.  Vectorised through a contingency table instead of the reference's
O(instances x clusters x N) mask loops; results are identical (tests/test_metrics.py).
"""
import numpy as np


def room_metrics(obj_id, cluster_label, with_sklearn=True):
    obj_id = np.asarray(obj_id)
    cluster_label = np.asarray(cluster_label)
    n_cluster = int(cluster_label.max()) if len(cluster_label) else 0
    unique_id, inv, count = np.unique(obj_id, return_inverse=True, return_counts=True)          # :325
    # contingency[g, j] = points of ground-truth instance g carrying cluster label j
    cont = np.zeros((len(unique_id), n_cluster + 1), dtype=np.int64)
    np.add.at(cont, (inv, cluster_label), 1)
    csize = cont.sum(axis=0)
    dt_match = np.zeros(n_cluster, dtype=bool)                                                  # :322
    cluster_label2 = np.zeros(len(cluster_label), dtype=int)
    gt_match = 0
    room_iou = []
    order = np.argsort(count)[::-1]                                                             # :327
    for k in range(len(unique_id)):
        g = order[k]
        best_iou = 0
        for j in range(1, n_cluster + 1):                                                       # :329
            if dt_match[j - 1]:
                continue
            inter = cont[g, j]
            iou = 1.0 * inter / (count[g] + csize[j] - inter)                                    # :331
            best_iou = max(best_iou, iou)
            if iou > 0.5:                                                                        # :333-337
                dt_match[j - 1] = True
                gt_match += 1
                cluster_label2[cluster_label == j] = k + 1
                break
        room_iou.append(best_iou)
    for j in range(1, n_cluster + 1):                                                            # :339-341
        if not dt_match[j - 1]:
            cluster_label2[cluster_label == j] = j + obj_id.max()
    out = dict(prc=float(np.mean(dt_match)) if n_cluster else float('nan'),                      # :342
               rcl=1.0 * gt_match / len(unique_id),                                              # :343
               iou=float(np.mean(room_iou)),                                                     # :344
               cluster_label2=cluster_label2)
    if with_sklearn:
        from sklearn.metrics import normalized_mutual_info_score, adjusted_rand_score, adjusted_mutual_info_score
        out['nmi'] = normalized_mutual_info_score(obj_id, cluster_label)                         # :346
        out['ami'] = adjusted_mutual_info_score(obj_id, cluster_label)                           # :347
        out['ars'] = adjusted_rand_score(obj_id, cluster_label)                                  # :348
    return out


def room_line(area, room_id, m):
    """The per-room line of test_region_grow.py:355."""
    return "Area %s room %d NMI: %.2f AMI: %.2f ARS: %.2f PRC: %.2f RCL: %.2f IOU: %.2f" % (
        str(area), room_id, m['nmi'], m['ami'], m['ars'], m['prc'], m['rcl'], m['iou'])


def aggregate_line(ms):
    """The final line of test_region_grow.py:379-381."""
    a = {k: np.array([m[k] for m in ms]) for k in ('nmi', 'ami', 'ars', 'prc', 'rcl', 'iou')}
    return 'NMI: %.2f+-%.2f AMI: %.2f+-%.2f ARS: %.2f+-%.2f PRC %.2f+-%.2f RCL %.2f+-%.2f IOU %.2f+-%.2f' % (
        a['nmi'].mean(), a['nmi'].std(), a['ami'].mean(), a['ami'].std(), a['ars'].mean(), a['ars'].std(),
        a['prc'].mean(), a['prc'].std(), a['rcl'].mean(), a['rcl'].std(), a['iou'].mean(), a['iou'].std())
