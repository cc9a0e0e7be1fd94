#!/usr/bin/env python3
"""Command-line driver with the job of the reference's ``test_region_grow.py`` / ``test_random_restart.py`` /
``test_beam_search.py``: rooms from an HDF5 file, weights from a TensorFlow checkpoint, region growing on the GPU, the
reference's per-region, per-room and aggregate lines and its timing table, optional PLY export.

    python region_grow.py --area 5                         # data/s3dis_area5.h5 + models/lrgnet_model5.ckpt (the
                                                           #   reference's layout, test_region_grow.py:68-98)
    python region_grow.py --h5 rooms.h5 --ckpt my/lrgnet.ckpt --restarts 10 --save out/
    python region_grow.py --h5 rooms.h5 --synthetic-weights --policy gt --timing
    python -m torch.distributed.run --nproc-per-node 8 region_grow.py --area 5 --gpus 8

Options of the reference that are kept: --area, --save, --resolution, --lite, --cross-domain/--train-area (model path
only).  ``--beam B`` selects the beam search of test_beam_search.py (B = BEAM_WIDTH, ``--search-width`` = SEARCH_WIDTH);
``--restarts R`` (R >= 2) the random-restart search of test_random_restart.py with ``--scoring np`` (its default) or ``ml``.

Random streams.  ``--rng counter`` (default): a counter-based stream keyed by (seed, room, seed point, restart, step) -- the
rooms of the file grow side by side on the GPU and results do not depend on batching or on the number of GPUs.
``--rng legacy``: NumPy's legacy generator consumed in the reference's order (choice, choice, random, random per step), masks
decided on the host from the GPU's logits.  By default every room gets its own ``RandomState(room index)``; the reference
script instead seeds once (``numpy.random.seed(0)``, test_region_grow.py:21) and draws all rooms of the file from that one
stream in file order -- ``--shared-stream`` does exactly that (one room at a time), which is the only way to reproduce the
reference's output beyond the first room.  Every room of the file is processed; the reference skips S3DIS rooms that are not
listed in ``data/s3dis_sampled.txt`` (test_region_grow.py:106-113): pass ``--room-list FILE`` with ``--room-names FILE`` to
apply the same filter.

Several GPUs (``--gpus N`` under torchrun, one process per GPU): the rooms are dealt to the ranks by equalised point count,
longest first (rooms are independent, test_region_grow.py:110-183); every rank grows its share; labels and metrics are
gathered over RCCL at the end (the only collective), rank 0 prints.
"""
import argparse
import os
import sys
import time

import numpy as np

CLASSES_S3DIS = ['clutter', 'board', 'bookcase', 'beam', 'chair', 'column', 'door', 'sofa', 'table', 'window', 'ceiling', 'floor',
                 'wall']          # class names are data (the reference's class_util.classes_s3dis); only the region lines use them


def parse(argv=None):
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
    ap.add_argument('--scoring', default='np', choices=['np', 'ml'],
                    help="restart score (test_random_restart.py:171-174): 'np' = points of the mask; 'ml' = accumulated log-likelihood of "
                         "the sampled masks, one scalar per restart (upstream's list reset at :194 breaks it from the second restart on)")
    ap.add_argument('--beam', type=int, default=0, help='beam width: > 0 selects the beam search of test_beam_search.py (--scoring np)')
    ap.add_argument('--search-width', type=int, default=3, help='children per beam entry (SEARCH_WIDTH)')
    ap.add_argument('--rng', default='counter', choices=['counter', 'legacy'])
    ap.add_argument('--shared-stream', action='store_true', help='--rng legacy with ONE RandomState(seed) for all rooms, in file order')
    ap.add_argument('--policy', default='net', choices=['net', 'gt', 'threshold'])
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--rooms-in-flight', type=int, default=68)
    ap.add_argument('--max-rooms', type=int, default=0)
    ap.add_argument('--room-names', default=None, help='one room name per line, in file order (data/<area>_room_name.txt of the reference)')
    ap.add_argument('--room-list', default=None, help='names of the rooms to process (data/s3dis_sampled.txt); needs --room-names')
    ap.add_argument('--device', default=None, help='default: cuda:<LOCAL_RANK>')
    ap.add_argument('--gpus', type=int, default=None, help='GPUs of this node, one process each: started by the script itself when it is not already '
                                                            'running under torch.distributed.run (whose LOCAL_WORLD_SIZE must agree when both are given; '
                                                            'without --gpus the launcher decides)')
    ap.add_argument('--speculate', type=int, default=-1,
                    help='regions of ONE room grown side by side (committed in seed order, voided and grown again on a conflict: identical labels); '
                         '-1 = automatic: learn_region_grow_amd.grow.auto_speculate(rooms in flight) -- several per room while the rooms on this GPU are '
                         'few (one room or scene per GPU is a single chain of dependent steps otherwise), 0 / 1 = off')
    ap.add_argument('--lanes', type=int, default=0, help='groups of slots on their own HIP streams (counter stream only); 0 = auto')
    ap.add_argument('--quiet-regions', action='store_true', help='do not print the per-region lines (test_region_grow.py:217)')
    ap.add_argument('--timing', action='store_true',
                    help="the reference's timing table (test_region_grow.py:382-390): rooms one at a time, HIP events round every launch")
    ap.add_argument('--preprocess', default='gpu-exact', choices=['gpu-exact', 'gpu-lapack', 'gpu', 'host'],
                    help="equalisation / normals / curvature (test_region_grow.py:119-173): 'gpu-exact' = all on the GPU (Jacobi eigen-solve), "
                         "the few points whose float32 features or seed-order position could differ under LAPACK redone with the reference's "
                         "numpy.linalg.svd on the host (features and seed order bit-identical, 4.5 x the rate of 'gpu-lapack'); 'gpu-lapack' = GPU "
                         "gathering and covariances + numpy.linalg.svd for every point (every output bit-identical); 'gpu' = all on the GPU "
                         "(features equal to float32 rounding, seed order up to near-ties); 'host' = vectorised NumPy")
    args = ap.parse_args(argv)
    if args.beam > 0:
        bad = [o for o, on in (('--rng legacy', args.rng != 'counter'), ('--restarts', args.restarts > 1), ('--lanes', args.lanes > 1),
                               ('--timing', args.timing), ('--scoring ml', args.scoring != 'np')) if on]
        if bad:
            ap.error('--beam drives the levels of test_beam_search.py itself; it does not combine with %s' % ', '.join(bad))
    if args.shared_stream and args.rng != 'legacy':
        ap.error('--shared-stream needs --rng legacy')
    if args.timing and (args.rng != 'counter' or args.restarts > 1):
        ap.error('--timing times the greedy loop under the counter stream')
    if args.scoring == 'ml' and (args.rng != 'counter' or args.restarts < 2):
        ap.error('--scoring ml needs --restarts R >= 2 and the counter stream')
    if args.room_list and not args.room_names:
        ap.error('--room-list needs --room-names')
    return args


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


def region_lines(room_id, res, obj_id, cls_id, classes):
    """The per-region lines of test_region_grow.py:217 (printed for regions above the cluster threshold only), from the
    device's region log and the final labels."""
    out = []
    obj_id = np.asarray(obj_id)
    cid = 0
    for reg in res.regions:
        if not reg['labeled']:
            continue
        cid += 1
        mask = res.cluster_label == cid
        target = int(obj_id[reg['seed']])
        gt = obj_id == target
        cname = classes[int(cls_id[np.nonzero(gt)[0][0]])] if classes is not None else ''
        iou = 1.0 * np.sum(np.logical_and(gt, mask)) / np.sum(np.logical_or(gt, mask))
        out.append('room %d target %3d %.4s: step %3d %4d/%4d points IOU %.3f add %.3f rmv %.3f %s' % (
            room_id, target, cname, reg['steps'], reg['points'], int(gt.sum()), iou, reg['add_acc'], reg['rmv_acc'], reg['reason']))
    return out


def timing_table(buckets):
    """The table of test_region_grow.py:382-390: mean +- std seconds per room (per iteration for iter_*) and share of the total."""
    keys = ['feature', 'net', 'neighbor', 'inlier', 'iter_net', 'iter_neighbor', 'iter_inlier']
    mean = {k: float(np.mean(buckets[k])) if len(buckets[k]) else 0.0 for k in keys}
    std = {k: float(np.std(buckets[k])) if len(buckets[k]) else 0.0 for k in keys}
    total = sum(mean.values())
    return ['%10s %6.2f+-%5.2fs %4.1f' % (k, mean[k], std[k], 100.0 * mean[k] / total if total else 0.0) for k in keys]


def main(argv=None):
    args = parse(argv)
    if (args.gpus or 1) > 1 and 'WORLD_SIZE' not in os.environ:
        # the launcher picks a free port itself (--standalone: a port bound and closed here could be taken before the ranks start)
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1', '--nproc-per-node',
                                  str(args.gpus), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv))
    import torch
    import torch.distributed as dist
    from learn_region_grow_amd import checkpoint, metrics, preprocess, preprocess_gpu, synthetic, dist as lrg_dist
    from learn_region_grow_amd import io as lio
    from learn_region_grow_amd.grow import LanedRegionGrower, RegionGrower
    from learn_region_grow_amd.lrgnet import LrgNetHIP

    if not torch.cuda.is_available():
        raise SystemExit('region_grow.py needs a GPU (the HIP path has no CPU fallback)')
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    if args.gpus is not None and local_world != args.gpus:
        raise SystemExit('region_grow.py: --gpus %d but LOCAL_WORLD_SIZE=%d' % (args.gpus, local_world))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    one_dev = os.environ.get('LRG_BENCH_ONE_DEVICE') == '1'      # testing on a 1-GPU box: every rank on cuda:0, collectives over gloo
    device = torch.device(args.device if args.device else 'cuda:%d' % (0 if one_dev else local))
    if one_dev and world > 1:
        # ranks sharing one chip: lock-step iterations -- a free-running launch needs all its workgroups resident at once, and two of them
        # side by side can each hold CUs the other waits for
        os.environ['LRG_FREE_RUN'] = '0'
    torch.cuda.set_device(device)                                # every launch below goes to this device's current stream
    coll_dev = device
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_dev:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            coll_dev = torch.device('cpu')
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    def say(*a):
        if rank == 0:
            print(*a)
            sys.stdout.flush()

    areas = args.area.split(',') if args.area else ['custom']
    all_metrics = []
    save_id = 0
    table = dict(feature=[], net=[], neighbor=[], inlier=[], iter_net=[], iter_neighbor=[], iter_inlier=[])
    for area in areas:
        if args.synthetic_weights:
            weights = synthetic.make_synthetic_weights(seed=args.seed, feature_size=args.feature_size, lite=args.lite or 0)
            say('Synthetic weights (seed %d)' % args.seed)
        else:
            mp = model_path(args, area)
            weights = checkpoint.load_lrgnet_weights(mp, feature_size=args.feature_size, lite=args.lite)
            say('Restored from %s' % mp)
        classes = None if 'kitti' in str(area) or area == 'scannet' else CLASSES_S3DIS
        net = LrgNetHIP(1, 1, 512, 512, args.feature_size, args.lite, device=device).load_weights(weights)
        all_points, all_obj_id, all_cls_id = lio.loadFromH5(data_path(args, area))
        room_ids = list(range(len(all_points)))
        if args.room_list:                                                                    # test_region_grow.py:101-113
            names = open(args.room_names).read().split('\n')
            keep = set(open(args.room_list).read().split('\n'))
            room_ids = [r for r in room_ids if '_'.join(names[r].split()) + '.h5' in keep]
        if args.max_rooms:
            room_ids = room_ids[:args.max_rooms]
        # ---- which rank grows which room: longest first by equalised point count ----
        sizes = [preprocess.equalized_count(all_points[r][:, :3], args.resolution) for r in room_ids] if world > 1 else [0] * len(room_ids)
        mine = lrg_dist.shard_rooms_lpt(sizes, world)[rank] if world > 1 else list(range(len(room_ids)))
        my_rooms = [room_ids[i] for i in mine]
        pre, t_feat = [], []
        for r in my_rooms:
            t0 = time.time()
            if args.preprocess == 'host':
                pre.append(preprocess.preprocess_room(all_points[r], all_obj_id[r], all_cls_id[r], resolution=args.resolution,
                                                      feature_size=args.feature_size))
            else:
                pre.append(preprocess_gpu.preprocess_room(all_points[r], all_obj_id[r], all_cls_id[r], resolution=args.resolution,
                                                          feature_size=args.feature_size, device=device,
                                                          eig={'gpu-lapack': 'lapack', 'gpu-exact': 'exact'}.get(args.preprocess, 'jacobi')))
            t_feat.append(time.time() - t0)
        rooms = [dict(points=p['points'], obj_id=p['obj_id'], order=p['order'].astype(np.int32), room_id=r) for r, p in zip(my_rooms, pre)]
        in_flight = max(1, min(args.rooms_in_flight, len(rooms)))
        kw = dict(rooms_in_flight=in_flight, restarts=max(1, args.restarts), rng=args.rng, seed=args.seed, policy=args.policy,
                  resolution=args.resolution)
        if args.scoring == 'ml':
            kw['scoring'] = 'ml'
        from learn_region_grow_amd.grow import auto_speculate
        spec_k = auto_speculate(in_flight) if args.speculate < 0 else args.speculate
        t0 = time.time()
        buckets = None
        if not rooms:
            results = []
        elif args.beam > 0:
            from learn_region_grow_amd.beam import BeamSearchGrower
            results = BeamSearchGrower(net, rooms_in_flight=in_flight, beam_width=args.beam, search_width=args.search_width, seed=args.seed,
                                       policy=args.policy, resolution=args.resolution).run(rooms)
        elif args.timing:
            kw['rooms_in_flight'] = 1
            results, buckets = RegionGrower(net, **kw).run_timed(rooms)
        elif args.shared_stream:
            kw['rooms_in_flight'] = 1
            results = RegionGrower(net, **kw).run(rooms, legacy_shared_seed=args.seed)
        else:
            results = None
            # (speculation only where no lane count was asked for: an explicit --lanes N > 1 means lock-step lanes)
            if (args.rng == 'counter' and max(1, args.restarts) == 1 and spec_k > 1 and args.lanes in (0, 1) and
                    RegionGrower.free_run_applies(net, rooms, in_flight * spec_k, **{k: v for k, v in kw.items() if k != 'rooms_in_flight'})):
                try:
                    results = RegionGrower(net, speculate=spec_k, **kw).run(rooms)
                except ValueError as e:      # (the host's check and the device's voxel words disagreed: the plain growers take the rooms)
                    say('speculation not applicable (%s): growing without it' % e)
                    results = None
            if results is None:
                if args.rng == 'counter' and args.lanes != 1:
                    results = LanedRegionGrower(net, lanes=args.lanes, **kw).run(rooms)
                else:
                    results = RegionGrower(net, **kw).run(rooms)
        t_grow = time.time() - t0
        # ---- per-room evaluation where the room was grown; lines, metrics and labels to rank 0 ----
        local_out = []
        for k, (r, res) in enumerate(zip(my_rooms, results)):
            m = metrics.room_metrics(pre[k]['obj_id'], res.filled_label)
            lines = [] if (args.quiet_regions or args.beam > 0) else region_lines(r, res, pre[k]['obj_id'], pre[k]['cls_id'], classes)
            local_out.append(dict(room=r, lines=lines, metrics={q: m[q] for q in ('nmi', 'ami', 'ars', 'prc', 'rcl', 'iou')},
                                  regions=len(res.regions), steps=res.total_steps, feature_s=t_feat[k],
                                  buckets=buckets[k] if buckets else None))
            if args.save:
                os.makedirs(args.save, exist_ok=True)
                cloud = np.array(all_points[r][:, :6], dtype=np.float64)
                colors = lio.label_colors(int(m['cluster_label2'].max()) + 1)              # test_region_grow.py:368-371
                cloud[:, 3:6] = colors[m['cluster_label2'], :][pre[k]['unequalized_idx']]
                name = ('scannet%d.ply' if area == 'scannet' else '%d.ply') % (save_id + room_ids.index(r))
                lio.savePLY(os.path.join(args.save, name), cloud)
        save_id += len(room_ids)
        if world > 1:
            # the final gather (the only collective): every room's labels to every rank over RCCL, the small per-room records as objects
            labels = lrg_dist.gather_room_labels(list(mine), [res.filled_label for res in results], len(room_ids), device=coll_dev)
            gathered = [None] * world
            dist.all_gather_object(gathered, local_out)
            local_out = [x for part in gathered for x in part]
            assert all(l is not None for l in labels)
            say('gathered the labels of %d rooms (%d points) from %d ranks' % (len(labels), sum(len(l) for l in labels), world))
        local_out.sort(key=lambda x: x['room'])
        for x in local_out:
            for ln in x['lines']:
                say(ln)
            say(metrics.room_line(area, x['room'], x['metrics']))
            all_metrics.append(x['metrics'])
            table['feature'].append(x['feature_s'])
            if x['buckets']:
                for q in ('net', 'neighbor', 'inlier'):
                    table[q].append(x['buckets'][q])
                    table['iter_' + q].extend(x['buckets']['iter_' + q])
        steps = sum(x['steps'] for x in local_out)
        say('%d rooms on %d GPU(s): preprocessing %.2f s (%s), region growing %.2f s on rank 0 (%d regions, %d grow steps, %.0f steps/s)' % (
            len(room_ids), world, sum(x['feature_s'] for x in local_out), args.preprocess, t_grow, sum(x['regions'] for x in local_out), steps,
            steps / max(t_grow, 1e-9)))
    if all_metrics:
        say(metrics.aggregate_line(all_metrics))
    if args.timing:
        for ln in timing_table(table):
            say(ln)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
