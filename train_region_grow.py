#!/usr/bin/env python3
"""Command-line driver with the job of the reference's ``train_region_grow.py``: staged tuples from ``data/staged_*.h5``
(written by ``stage_data.py`` or by ``learn_region_grow_amd.stage``), LrgNet trained on the GPU with the reference's losses
and Adam (learn_region_grow_util.py:165-189), the reference's epoch lines, weights saved as a TensorFlow checkpoint bundle
that ``region_grow.py`` (and the reference's ``Saver.restore``) reads.

    python train_region_grow.py --train-area 1,2,3,4,6 --val-area 5            # data/multiseed/seed<e % 8>_area<k>.h5 (training areas, one seed
                                                                               # per epoch), data/staged_area5.h5 (validation) -> models/lrgnet_model5.ckpt
    python train_region_grow.py --multiseed 0 --train-area 1,2,3,4,6           # data/staged_area<k>.h5
    python train_region_grow.py --staged my_tuples.h5 --ckpt out/lrgnet.ckpt --epochs 20

Options of the reference that are kept: --train-area, --val-area, --lite, --multiseed (data/multiseed/seed<k>_area<a>.h5, one seed
per epoch, train_region_grow.py:69-76), --cross-domain.  Batch assembly (pad / subsample each tuple to 512 + 512 points with the
legacy NumPy generator seeded 0) follows train_region_grow.py:156-175 call for call.
"""
import argparse
import os
import sys
import time

import numpy as np


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--train-area', default='1,2,3,4,6')
    ap.add_argument('--val-area', default=None)
    ap.add_argument('--staged', default=None, help='comma list of staged files (overrides the area naming scheme)')
    ap.add_argument('--data-dir', default='data')
    ap.add_argument('--model-dir', default='models')
    ap.add_argument('--ckpt', default=None, help='output checkpoint prefix (default: the reference naming scheme)')
    ap.add_argument('--cross-domain', action='store_true')
    ap.add_argument('--multiseed', type=int, default=8, help='seeds cycled per epoch, data/multiseed/seed<k>_area<a>.h5 (reference default 8; 0 = data/staged_area<a>.h5)')
    ap.add_argument('--lite', type=int, default=None)
    ap.add_argument('--feature-size', type=int, default=13, choices=[6, 9, 12, 13])
    ap.add_argument('--batch-size', type=int, default=100)
    ap.add_argument('--epochs', type=int, default=50)
    ap.add_argument('--val-step', type=int, default=7)
    ap.add_argument('--learning-rate', type=float, default=1e-3)
    ap.add_argument('--init', default=None, help='checkpoint prefix to start from (default: the reference initialiser)')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--device', default='cuda:0')
    return ap.parse_args(argv)


def model_path(args):
    """train_region_grow.py:37-53."""
    if args.ckpt:
        return args.ckpt
    val = args.val_area.split(',')[0] if args.val_area else 'None'
    d = args.model_dir
    if args.cross_domain:
        return os.path.join(d, 'cross_domain', 'lrgnet_%s.ckpt' % args.train_area.split(',')[0])
    suffix = {6: '_xyz', 9: '_xyzrgb', 12: '_xyzrgbn'}.get(args.feature_size, '')
    if not suffix and args.lite is not None:
        suffix = '_lite_%d' % args.lite
    return os.path.join(d, 'lrgnet_model%s%s.ckpt' % (val, suffix))


def staged_path(args, area, epoch, is_train=True):
    """train_region_grow.py:69-78: the multiseed files are for the TRAINING areas only (`elif MULTISEED > 0 and AREA in TRAIN_AREA`);
    a validation area always reads data/staged_area<a>.h5."""
    if str(area).startswith('synthetic'):
        return os.path.join(args.data_dir, 'staged_%s.h5' % area)
    if args.multiseed > 0 and is_train:
        return os.path.join(args.data_dir, 'multiseed', 'seed%d_area%s.h5' % (epoch % args.multiseed, area))
    return os.path.join(args.data_dir, 'staged_area%s.h5' % area)


def run_epoch(trainer, data, rs, batch_size, train=True, shuffle=True):
    """One pass over the tuples in batches (train_region_grow.py:141-183 / :186-219): mean loss and add / remove precision, recall."""
    from learn_region_grow_amd.stage import assemble_batch
    idx = np.arange(len(data['points']))
    if shuffle:
        rs.shuffle(idx)
    rows = []
    for b in range(len(idx) // batch_size):
        xi, xn, ia, ir = assemble_batch(data, idx[b * batch_size:(b + 1) * batch_size], rs, batch_size, trainer.Ni, trainer.Nn)
        if train:
            rows.append(trainer.train_step(xi, xn, ia, ir))
        else:
            sc = trainer.evaluate(xi, xn, ia, ir)              # forward + loss only (no dW / dX products)
            rows.append((sc['loss'], sc['add_prc'], sc['add_rcl'], sc['remove_prc'], sc['remove_rcl']))
    return np.mean(np.array(rows, dtype=np.float64), axis=0) if rows else np.zeros(5)


def main(argv=None):
    args = parse(argv)
    import torch
    from learn_region_grow_amd import checkpoint, stage, synthetic
    from learn_region_grow_amd.train import LrgNetTrainer
    if not torch.cuda.is_available():
        raise SystemExit('train_region_grow.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(torch.device(args.device))
    rs = np.random.RandomState(args.seed)                                         # numpy.random.seed(0), train_region_grow.py:17
    trainer = LrgNetTrainer(args.batch_size, 512, 512, args.feature_size, args.lite, device=args.device, learning_rate=args.learning_rate)
    if args.init:
        trainer.load_weights(checkpoint.load_lrgnet_weights(args.init, feature_size=args.feature_size, lite=args.lite))
    else:
        trainer.load_weights(synthetic.make_reference_init_weights(seed=args.seed, feature_size=args.feature_size, lite=args.lite or 0))
    train_areas = args.train_area.split(',')
    val_areas = args.val_area.split(',') if args.val_area else []
    loaded, data, val = None, None, None
    epoch_time = []
    for epoch in range(args.epochs):
        files = args.staged.split(',') if args.staged else [staged_path(args, a, epoch) for a in train_areas]
        if loaded != files:                                                      # reloaded per epoch only under --multiseed (:60)
            parts = []
            for f in files:
                if os.path.exists(f):
                    print('Loading %s ...' % f)
                    parts.append(stage.load_staged(f, args.feature_size))
            if not parts:
                # (the reference skips a missing multiseed file, :75-76, and would then fail on an empty training set)
                print('warning: epoch %d: none of %s exists -- nothing to train on' % (epoch, ', '.join(files)), file=sys.stderr)
                continue
            data = {k: [x for p in parts for x in p[k]] for k in ('points', 'remove', 'neighbor_points', 'add')}
            loaded = files
            print('train', len(data['points']), data['points'][0].shape, len(data['neighbor_points']))
        t0 = time.time()
        ls, ap, ar, rp, rr = run_epoch(trainer, data, rs, args.batch_size)
        epoch_time.append(time.time() - t0)
        print('Epoch %d loss %.2f add %.2f/%.2f rmv %.2f/%.2f' % (epoch, ls, ap, ar, rp, rr))          # :183
        if val_areas and epoch % args.val_step == args.val_step - 1:                                   # :185-219
            if val is None:
                vfiles = [staged_path(args, a, 0, is_train=False) for a in val_areas]
                for f in vfiles:
                    if not os.path.exists(f):
                        print('warning: validation file %s does not exist' % f, file=sys.stderr)
                vparts = [stage.load_staged(f, args.feature_size) for f in vfiles if os.path.exists(f)]
                val = {k: [x for p in vparts for x in p[k]] for k in ('points', 'remove', 'neighbor_points', 'add')} if vparts else None
            if val:
                ls, ap, ar, rp, rr = run_epoch(trainer, val, rs, args.batch_size, train=False, shuffle=False)
                print('Validation %d loss %.2f add %.2f/%.2f rmv %.2f/%.2f' % (epoch, ls, ap, ar, rp, rr))
    if not epoch_time:
        raise SystemExit('no staged training file was found in any epoch (looked for e.g. %s): nothing trained, no checkpoint written'
                         % (args.staged or staged_path(args, train_areas[0], 0)))
    print('Avg Epoch Time: %.3f' % np.mean(epoch_time))
    out = model_path(args)
    os.makedirs(os.path.dirname(out) or '.', exist_ok=True)
    checkpoint.write_bundle(out, trainer.checkpoint_numpy())      # variables + Adam slots + beta powers + global step (99 entries)
    print('Saved %s' % out)
    return 0


if __name__ == '__main__':
    sys.exit(main())
