"""Oracle: per-room metric block (test infrastructure, see oracle/__init__.py).

Restates /root/reference/test_region_grow.py:319-355: greedy IoU>0.5 matching of
ground-truth instances (descending size) to predicted clusters, PRC / RCL / mean best IoU,
plus sklearn NMI / AMI / ARS.
"""
import numpy as np


def room_metrics(obj_id, cluster_label, with_sklearn=True):
    obj_id = np.asarray(obj_id)
    cluster_label = np.asarray(cluster_label)
    gt_match = 0
    dt_match = np.zeros(cluster_label.max(), dtype=bool)                       # :322
    cluster_label2 = np.zeros(len(cluster_label), dtype=int)
    room_iou = []
    unique_id, count = np.unique(obj_id, return_counts=True)                   # :325
    for k in range(len(unique_id)):
        i = unique_id[np.argsort(count)][::-1][k]                              # :327
        best_iou = 0
        for j in range(1, cluster_label.max() + 1):
            if not dt_match[j - 1]:
                iou = 1.0 * np.sum(np.logical_and(obj_id == i, cluster_label == j)) / np.sum(np.logical_or(obj_id == i, cluster_label == j))
                best_iou = max(best_iou, iou)
                if iou > 0.5:
                    dt_match[j - 1] = True
                    gt_match += 1
                    cluster_label2[cluster_label == j] = k + 1
                    break
        room_iou.append(best_iou)
    for j in range(1, cluster_label.max() + 1):                                # :339-341
        if not dt_match[j - 1]:
            cluster_label2[cluster_label == j] = j + obj_id.max()
    out = dict(prc=float(np.mean(dt_match)) if len(dt_match) else float('nan'),    # :342
               rcl=1.0 * gt_match / len(set(obj_id.tolist())),                  # :343
               iou=float(np.mean(room_iou)),                                    # :344
               cluster_label2=cluster_label2)
    if with_sklearn:
        from sklearn.metrics import normalized_mutual_info_score, adjusted_rand_score, adjusted_mutual_info_score
        out['nmi'] = normalized_mutual_info_score(obj_id, cluster_label)        # :346
        out['ami'] = adjusted_mutual_info_score(obj_id, cluster_label)          # :347
        out['ars'] = adjusted_rand_score(obj_id, cluster_label)                 # :348
    return out
