#!/usr/bin/env python3
"""`odtk` command line (reference odtk/main.py): `train`, `infer`, `export` with the reference's flags.

What differs from the reference, and why:

  * one process per GPU either way, but both launch styles work: run plainly, the command spawns a worker per
    visible GPU like the reference (main.py:243-251); run under `torchrun` (RANK / WORLD_SIZE / LOCAL_RANK in
    the environment), each process is one worker and nothing is spawned.  `backend="nccl"` is RCCL on ROCm.
  * every worker loads the checkpoint itself instead of receiving a pickled, shared-memory model from the
    parent (main.py:244-245): no 150 MB pickle per rank, no CPU shared-memory segment behind GPU-resident
    weights.
  * `--with-apex` / `--with-dali` and TensorRT engines (`.plan` / `.engine`, `export` to anything) name
    dependencies the north star drops; they are accepted by the parser and refused with a clear error.
"""
import argparse
import os
import sys

import torch
import torch.cuda
import torch.distributed
import torch.multiprocessing

from . import infer, train
from .model import Model


def _flag(name, kind=None, default=None, text='', **extra):
    """One row of the flag tables below: (name, argparse keyword arguments)."""
    spec = dict(extra, help=text)
    if kind is bool:
        spec['action'] = 'store_true'
    else:
        spec.update(type=kind, default=default)
    return name, spec


_DROPPED = 'names a dependency this build drops; refused at run time'
_SIZES = [_flag('--batch', int, None, 'images per step over all GPUs (default: 2 per GPU)', metavar='size'),
          _flag('--resize', int, 800, 'short side after resizing', metavar='scale'),
          _flag('--max-size', int, 1333, 'cap on the long side after resizing', metavar='max')]
_COMMON = [_flag('--with-apex', bool, text=_DROPPED), _flag('--with-dali', bool, text=_DROPPED),
           _flag('--workers', int, 8, 'loader processes per GPU (the reference hard-codes 2; ~6 feed one MI355X)', metavar='num')]

# same names, types and defaults as the reference's parser (main.py:15-118)
TRAIN_FLAGS = [
    _flag('--annotations', str, None, 'COCO-style annotation file', metavar='path', required=True),
    _flag('--images', str, '.', 'image directory', metavar='path'),
    _flag('--backbone', str, ['ResNet50FPN'], 'one backbone or several', nargs='+'),
    _flag('--classes', int, 80, 'number of object classes', metavar='num'),
    *_SIZES,
    _flag('--jitter', int, [640, 1024], 'range of short-side sizes drawn per image', nargs=2, metavar='min max'),
    _flag('--iters', int, 90000, 'training iterations', metavar='number'),
    _flag('--milestones', int, [60000, 80000], 'iterations at which the learning rate drops', nargs='*'),
    _flag('--schedule', float, 1, 'stretch factor for --iters and --milestones', metavar='scale'),
    _flag('--full-precision', bool, text='fp32 instead of mixed precision'),
    _flag('--lr', float, 0.01, 'peak learning rate', metavar='value'),
    _flag('--warmup', int, 1000, 'iterations of linear warm-up', metavar='iterations'),
    _flag('--gamma', float, 0.1, 'learning-rate factor at a milestone', metavar='value'),
    _flag('--override', bool, text='start from scratch even if the model file exists'),
    _flag('--val-annotations', str, None, 'annotation file of the validation set', metavar='path'),
    _flag('--val-images', str, None, 'image directory of the validation set', metavar='path'),
    _flag('--post-metrics', str, None, 'POST training metrics to this address', metavar='url'),
    _flag('--fine-tune', str, None, 'initialise from this checkpoint', metavar='path'),
    _flag('--logdir', str, None, 'directory for scalar logs', metavar='logdir'),
    _flag('--val-iters', int, 8000, 'iterations between two validations', metavar='number'),
    *_COMMON,
    _flag('--augment-rotate', bool, text='random quarter turns'),
    _flag('--augment-free-rotate', float, [0, 0], 'accepted for compatibility (unused by the reference as well)', nargs=2,
          metavar='value value'),
    _flag('--augment-brightness', float, 0.002, 'sigma of the brightness factor', metavar='value'),
    _flag('--augment-contrast', float, 0.002, 'sigma of the contrast factor', metavar='value'),
    _flag('--augment-hue', float, 0.0002, 'sigma of the hue shift', metavar='value'),
    _flag('--augment-saturation', float, 0.002, 'sigma of the saturation factor', metavar='value'),
    _flag('--regularization-l2', float, 0.0001, 'weight decay', metavar='value'),
    _flag('--rotated-bbox', bool, text='boxes are [x, y, w, h, theta]'),
    _flag('--anchor-ious', float, [0.4, 0.5], 'background / foreground overlap thresholds', nargs=2, metavar='value value'),
    _flag('--absolute-angle', bool, text='regress the absolute angle instead of -45..45 degrees'),
]
INFER_FLAGS = [
    _flag('--images', str, '.', 'image directory', metavar='path'),
    _flag('--annotations', str, None, 'annotation file: evaluate against it', metavar='annotations'),
    _flag('--output', str, ['detections.json'], 'JSON file(s) for the detections', nargs='+', metavar='file'),
    *_SIZES, *_COMMON,
    _flag('--full-precision', bool, text='fp32 instead of mixed precision'),
    _flag('--rotated-bbox', bool, text='the model predicts rotated boxes'),
]
EXPORT_FLAGS = [                                                   # parsed for compatibility; `export` itself is refused
    _flag('--size', int, [1280], nargs='+', metavar='height width'), _flag('--full-precision', bool), _flag('--int8', bool),
    _flag('--calibration-batches', int, 2, metavar='size'), _flag('--calibration-images', str, '', metavar='path'),
    _flag('--calibration-table', str, '', metavar='path'), _flag('--verbose', bool), _flag('--rotated-bbox', bool),
    _flag('--dynamic-batch-opts', int, [1, 8, 16], nargs=3, metavar='value value value'),
]


def parse(args):
    parser = argparse.ArgumentParser(description='ODTK: Object Detection Toolkit.')
    parser.add_argument('--master', metavar='address:port', type=str, default='127.0.0.1:29500',
                        help='rendezvous of the per-GPU workers')
    commands = parser.add_subparsers(help='sub-command', dest='command')
    commands.required = True
    per_gpu = 2 * max(1, torch.cuda.device_count())
    for name, positional, flags, text in (
            ('train', [('model', 'checkpoint to write (and to resume from when it exists)')], TRAIN_FLAGS, 'train a network'),
            ('infer', [('model', 'checkpoint to load')], INFER_FLAGS, 'run inference'),
            ('export', [('model', 'checkpoint to load'), ('export', 'output file')], EXPORT_FLAGS,
             'TensorRT export (dropped: refused)')):
        sub = commands.add_parser(name, help=text)
        for arg, arg_text in positional:
            sub.add_argument(arg, type=str, help=arg_text)
        for flag, spec in flags:
            spec = dict(spec)
            if flag == '--batch':
                spec['default'] = per_gpu
            sub.add_argument(flag, **spec)
    return parser.parse_args(args)


def load_model(args, verbose=False):
    """Fresh model (train on a new path, or --override), or a `.pth` / `.torch` checkpoint with its training
    state (reference main.py:121-152).  TensorRT plans are refused."""
    if args.command != 'train' and not os.path.isfile(args.model):
        raise RuntimeError('Model file {} does not exist!'.format(args.model))
    state = {}
    ext = os.path.splitext(args.model)[1]
    if args.command == 'train' and (not os.path.exists(args.model) or args.override):
        if verbose:
            print('Initializing model...')
        model = Model(backbones=args.backbone, classes=args.classes, rotated_bbox=args.rotated_bbox,
                      anchor_ious=args.anchor_ious)
        model.initialize(args.fine_tune)
    elif ext in ('.pth', '.torch'):
        if verbose:
            print('Loading model from {}...'.format(os.path.basename(args.model)))
        model, state = Model.load(filename=args.model, rotated_bbox=args.rotated_bbox)
    elif ext in ('.engine', '.plan'):
        raise RuntimeError('TensorRT engines are not supported on MI355X: "{}"'.format(args.model))
    else:
        raise RuntimeError('Invalid model format "{}"!'.format(ext))
    model.freeze_unused_params()                                   # (reference: unused params are frozen, :136,143)
    if verbose:
        print(model)
    state['path'] = args.model
    return model, state


def launched_by_torchrun():
    return 'RANK' in os.environ and 'WORLD_SIZE' in os.environ


def worker(rank, args, world, spawned=False):
    """One process, one GPU (or the CPU)."""
    if spawned:
        address, port = args.master.rsplit(':', 1)
        os.environ.update({'MASTER_ADDR': address, 'MASTER_PORT': port, 'WORLD_SIZE': str(world), 'RANK': str(rank)})
    if torch.cuda.is_available():
        local = int(os.environ.get('LOCAL_RANK', rank))
        if local >= torch.cuda.device_count():
            raise RuntimeError('rank %d (local rank %d) has no GPU: this host has %d -- one process per GPU' % (
                rank, local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
    if world > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group(backend='nccl' if torch.cuda.is_available() else 'gloo',
                                             init_method='env://', world_size=world, rank=rank)
    if args.command != 'export' and args.batch % world != 0:
        raise RuntimeError('Batch size should be a multiple of the number of GPUs')

    model, state = load_model(args, verbose=(rank == 0))
    if model.angles is not None:
        args.rotated_bbox = True
    try:
        if args.command == 'train':
            return train.train(model, state, args.images, args.annotations, args.val_images or args.images,
                               args.val_annotations, args.resize, args.max_size, args.jitter, args.batch,
                               int(args.iters * args.schedule), args.val_iters, args.lr, args.warmup,
                               [int(m * args.schedule) for m in args.milestones], args.gamma, rank, world=world,
                               mixed_precision=not args.full_precision, with_apex=args.with_apex, use_dali=args.with_dali,
                               metrics_url=args.post_metrics, logdir=args.logdir, verbose=(rank == 0),
                               rotate_augment=args.augment_rotate, augment_brightness=args.augment_brightness,
                               augment_contrast=args.augment_contrast, augment_hue=args.augment_hue,
                               augment_saturation=args.augment_saturation, regularization_l2=args.regularization_l2,
                               rotated_bbox=args.rotated_bbox, absolute_angle=args.absolute_angle,
                               num_workers=args.workers)
        if args.command == 'infer':
            return infer.infer(model, args.images, args.output, args.resize, args.max_size, args.batch,
                               annotations=args.annotations, mixed_precision=not args.full_precision,
                               is_master=(rank == 0), world=world, with_apex=args.with_apex, use_dali=args.with_dali,
                               verbose=(rank == 0), rotated_bbox=args.rotated_bbox, num_workers=args.workers)
        return model.export(args.size, args.dynamic_batch_opts)     # raises: TensorRT is dropped
    finally:
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


def main(args=None):
    """Entry point for the odtk command."""
    args = parse(args or sys.argv[1:])
    if launched_by_torchrun():
        return worker(int(os.environ['RANK']), args, int(os.environ['WORLD_SIZE']))
    world = torch.cuda.device_count()
    if args.command == 'export' or world <= 1:
        return worker(0, args, 1)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs on this driver
    torch.multiprocessing.spawn(worker, args=(args, world, True), nprocs=world)


if __name__ == '__main__':
    main()
