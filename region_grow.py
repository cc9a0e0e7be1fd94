#!/usr/bin/env python3
"""Command-line driver with the job of the reference's ``test_region_grow.py`` / ``test_random_restart.py``: rooms from an
HDF5 file, weights from a TensorFlow checkpoint, region growing on the GPU, the reference's per-room and aggregate metric
lines, optional PLY export.

    python region_grow.py --area 5                         # data/s3dis_area5.h5 + models/lrgnet_model5.ckpt (the
                                                           #   reference's layout, test_region_grow.py:68-98)
    python region_grow.py --h5 rooms.h5 --ckpt my/lrgnet.ckpt --restarts 10 --save out/
    python region_grow.py --h5 rooms.h5 --synthetic-weights --policy gt

Options of the reference that are kept: --area, --save, --resolution, --lite, --cross-domain/--train-area (model path
only).  ``--beam B`` selects the beam search of test_beam_search.py (B = BEAM_WIDTH, ``--search-width`` = SEARCH_WIDTH, scoring np);
``--restarts R`` (R >= 2) selects the random-restart search of test_random_restart.py with its default
``--scoring np`` (its ``ml`` scoring raises at the second restart, test_random_restart.py:194 / :269, and is not offered).
``--rng legacy`` reproduces the reference's per-room NumPy random stream (one room at a time on the host side of the loop);
the default ``counter`` batches every room of the file on the GPU.
"""
import argparse
import os
import sys
import time

import numpy as np


def parse():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--area', default=None, help="comma list; 'scannet', 's3dis', 'kitti_train', 'kitti_val' or an S3DIS area number")
    ap.add_argument('--h5', default=None, help='room file (overrides --area for the data)')
    ap.add_argument('--ckpt', default=None, help='checkpoint prefix (overrides the reference naming scheme)')
    ap.add_argument('--synthetic-weights', action='store_true', help='seeded random weights instead of a checkpoint')
    ap.add_argument('--data-dir', default='data')
    ap.add_argument('--model-dir', default='models')
    ap.add_argument('--save', nargs='?', const='data/results/lrg', default=None, help='write <dir>/<n>.ply per room')
    ap.add_argument('--cross-domain', action='store_true')
    ap.add_argument('--train-area', default=None)
    ap.add_argument('--resolution', type=float, default=0.1)
    ap.add_argument('--lite', type=int, default=None)
    ap.add_argument('--feature-size', type=int, default=13, choices=[6, 9, 12, 13])
    ap.add_argument('--restarts', type=int, default=1)
    ap.add_argument('--beam', type=int, default=0, help='beam width: > 0 selects the beam search of test_beam_search.py (--scoring np)')
    ap.add_argument('--search-width', type=int, default=3, help='children per beam entry (SEARCH_WIDTH)')
    ap.add_argument('--rng', default='counter', choices=['counter', 'legacy'])
    ap.add_argument('--policy', default='net', choices=['net', 'gt', 'threshold'])
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--rooms-in-flight', type=int, default=68)
    ap.add_argument('--max-rooms', type=int, default=0)
    ap.add_argument('--device', default='cuda:0')
    ap.add_argument('--lanes', type=int, default=0, help='half-batches on their own HIP streams (counter stream only); 0 = auto')
    ap.add_argument('--preprocess', default='gpu-lapack', choices=['gpu-lapack', 'gpu', 'host'],
                    help="equalisation / normals / curvature (test_region_grow.py:119-173): 'gpu-lapack' = GPU gathering and "
                         "covariances + the reference's numpy.linalg.svd on the host (bit-identical features); 'gpu' = all on the "
                         "GPU (Jacobi eigen-solve, features equal to float32 rounding); 'host' = vectorised NumPy")
    return ap.parse_args()


def model_path(args, area):
    """test_region_grow.py:70-87."""
    if args.ckpt:
        return args.ckpt
    d = args.model_dir
    if args.cross_domain:
        return os.path.join(d, 'cross_domain', 'lrgnet_%s.ckpt' % args.train_area)
    suffix = {6: '_xyz', 9: '_xyzrgb', 12: '_xyzrgbn'}.get(args.feature_size, '')
    if not suffix and args.lite is not None:
        suffix = '_lite_%d' % args.lite
    return os.path.join(d, 'lrgnet_model%s%s.ckpt' % (area, suffix))


def data_path(args, area):
    """test_region_grow.py:95-98."""
    if args.h5:
        return args.h5
    if area in ('scannet', 's3dis', 'kitti_train', 'kitti_val'):
        return os.path.join(args.data_dir, '%s.h5' % area)
    return os.path.join(args.data_dir, 's3dis_area%s.h5' % area)


def main():
    args = parse()
    import torch
    from learn_region_grow_amd import checkpoint, metrics, preprocess, preprocess_gpu, synthetic
    from learn_region_grow_amd import io as lio
    from learn_region_grow_amd.grow import LanedRegionGrower, RegionGrower
    from learn_region_grow_amd.lrgnet import LrgNetHIP

    if not torch.cuda.is_available():
        raise SystemExit('region_grow.py needs a GPU (the HIP path has no CPU fallback)')
    areas = args.area.split(',') if args.area else ['custom']
    all_metrics = []
    save_id = 0
    for area in areas:
        if args.synthetic_weights:
            weights = synthetic.make_synthetic_weights(seed=args.seed, feature_size=args.feature_size, lite=args.lite or 0)
            print('Synthetic weights (seed %d)' % args.seed)
        else:
            mp = model_path(args, area)
            weights = checkpoint.load_lrgnet_weights(mp, feature_size=args.feature_size, lite=args.lite)
            print('Restored from %s' % mp)
        net = LrgNetHIP(1, 1, 512, 512, args.feature_size, args.lite, device=args.device).load_weights(weights)
        all_points, all_obj_id, all_cls_id = lio.loadFromH5(data_path(args, area))
        n_rooms = len(all_points) if not args.max_rooms else min(args.max_rooms, len(all_points))
        t0 = time.time()
        if args.preprocess == 'host':
            pre = [preprocess.preprocess_room(all_points[r], all_obj_id[r], all_cls_id[r], resolution=args.resolution,
                                              feature_size=args.feature_size) for r in range(n_rooms)]
        else:
            pre = [preprocess_gpu.preprocess_room(all_points[r], all_obj_id[r], all_cls_id[r], resolution=args.resolution,
                                                  feature_size=args.feature_size, device=args.device,
                                                  eig='lapack' if args.preprocess == 'gpu-lapack' else 'jacobi')
                   for r in range(n_rooms)]
        t_feature = time.time() - t0
        rooms = [dict(points=p['points'], obj_id=p['obj_id'], order=p['order'].astype(np.int32), room_id=r)
                 for r, p in enumerate(pre)]
        kw = dict(rooms_in_flight=min(args.rooms_in_flight, n_rooms), restarts=max(1, args.restarts), rng=args.rng,
                  seed=args.seed, policy=args.policy, resolution=args.resolution)
        if args.beam > 0:
            from learn_region_grow_amd.beam import BeamSearchGrower
            gr = BeamSearchGrower(net, rooms_in_flight=min(args.rooms_in_flight, n_rooms), beam_width=args.beam,
                                  search_width=args.search_width, seed=args.seed, policy=args.policy, resolution=args.resolution)
        elif args.rng == 'counter' and args.lanes != 1:
            gr = LanedRegionGrower(net, lanes=args.lanes, **kw)
        else:
            gr = RegionGrower(net, **kw)
        t0 = time.time()
        results = gr.run(rooms)
        t_grow = time.time() - t0
        steps = sum(reg['steps'] for res in results for reg in res.regions)
        for r, res in enumerate(results):
            m = metrics.room_metrics(pre[r]['obj_id'], res.filled_label)
            all_metrics.append(m)
            print(metrics.room_line(area, r, m))
            if args.save:
                os.makedirs(args.save, exist_ok=True)
                cloud = np.array(all_points[r][:, :6], dtype=np.float64)
                colors = lio.label_colors(int(m['cluster_label2'].max()) + 1)              # test_region_grow.py:368-371
                cloud[:, 3:6] = colors[m['cluster_label2'], :][pre[r]['unequalized_idx']]
                name = ('scannet%d.ply' if area == 'scannet' else '%d.ply') % save_id
                lio.savePLY(os.path.join(args.save, name), cloud)
                save_id += 1
        print('%d rooms: preprocessing %.2f s (%s), region growing %.2f s (%d regions, %d grow steps, %.0f steps/s)' % (
            n_rooms, t_feature, args.preprocess, t_grow, sum(len(res.regions) for res in results), steps, steps / max(t_grow, 1e-9)))
    print(metrics.aggregate_line(all_metrics))
    return 0


if __name__ == '__main__':
    sys.exit(main())
