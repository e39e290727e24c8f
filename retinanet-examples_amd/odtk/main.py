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


def parse(args):
    parser = argparse.ArgumentParser(description='ODTK: Object Detection Toolkit.')
    parser.add_argument('--master', metavar='address:port', type=str, help='Address and port of the master worker',
                        default='127.0.0.1:29500')
    subparsers = parser.add_subparsers(help='sub-command', dest='command')
    subparsers.required = True
    devcount = max(1, torch.cuda.device_count())

    p = subparsers.add_parser('train', help='train a network')
    p.add_argument('model', type=str, help='path to output model or checkpoint to resume from')
    p.add_argument('--annotations', metavar='path', type=str, help='path to COCO style annotations', required=True)
    p.add_argument('--images', metavar='path', type=str, help='path to images', default='.')
    p.add_argument('--backbone', action='store', type=str, nargs='+', help='backbone model (or list of)',
                   default=['ResNet50FPN'])
    p.add_argument('--classes', metavar='num', type=int, help='number of classes', default=80)
    p.add_argument('--batch', metavar='size', type=int, help='batch size', default=2 * devcount)
    p.add_argument('--resize', metavar='scale', type=int, help='resize to given size', default=800)
    p.add_argument('--max-size', metavar='max', type=int, help='maximum resizing size', default=1333)
    p.add_argument('--jitter', metavar='min max', type=int, nargs=2, help='jitter size within range', default=[640, 1024])
    p.add_argument('--iters', metavar='number', type=int, help='number of iterations to train for', default=90000)
    p.add_argument('--milestones', action='store', type=int, nargs='*',
                   help='list of iteration indices where learning rate decays', default=[60000, 80000])
    p.add_argument('--schedule', metavar='scale', type=float, help='scale schedule (affecting iters and milestones)',
                   default=1)
    p.add_argument('--full-precision', help='train in full precision', action='store_true')
    p.add_argument('--lr', metavar='value', help='learning rate', type=float, default=0.01)
    p.add_argument('--warmup', metavar='iterations', help='numer of warmup iterations', type=int, default=1000)
    p.add_argument('--gamma', metavar='value', type=float, help='multiplicative factor of learning rate decay',
                   default=0.1)
    p.add_argument('--override', help='override model', action='store_true')
    p.add_argument('--val-annotations', metavar='path', type=str, help='path to COCO style validation annotations')
    p.add_argument('--val-images', metavar='path', type=str, help='path to validation images')
    p.add_argument('--post-metrics', metavar='url', type=str, help='post metrics to specified url')
    p.add_argument('--fine-tune', metavar='path', type=str, help='fine tune a pretrained model')
    p.add_argument('--logdir', metavar='logdir', type=str, help='directory where to write logs')
    p.add_argument('--val-iters', metavar='number', type=int, help='number of iterations between each validation',
                   default=8000)
    p.add_argument('--with-apex', help='(dropped dependency: refused)', action='store_true')
    p.add_argument('--with-dali', help='(dropped dependency: refused)', action='store_true')
    p.add_argument('--augment-rotate', help='use four-fold rotational augmentation', action='store_true')
    p.add_argument('--augment-free-rotate', type=float, metavar='value value', nargs=2, default=[0, 0],
                   help='rotate images by an arbitrary angle, between min and max (in degrees)')
    p.add_argument('--augment-brightness', metavar='value', type=float, help='adjust the brightness of the image.',
                   default=0.002)
    p.add_argument('--augment-contrast', metavar='value', type=float, help='adjust the contrast of the image.',
                   default=0.002)
    p.add_argument('--augment-hue', metavar='value', type=float, help='adjust the hue of the image.', default=0.0002)
    p.add_argument('--augment-saturation', metavar='value', type=float, help='adjust the saturation of the image.',
                   default=0.002)
    p.add_argument('--regularization-l2', metavar='value', type=float, help='L2 regularization for optim',
                   default=0.0001)
    p.add_argument('--rotated-bbox', help='detect rotated bounding boxes [x, y, w, h, theta]', action='store_true')
    p.add_argument('--anchor-ious', metavar='value value', type=float, nargs=2, help='anchor/bbox overlap threshold',
                   default=[0.4, 0.5])
    p.add_argument('--absolute-angle', help='regress absolute angle (rather than -45 to 45 degrees.',
                   action='store_true')
    p.add_argument('--workers', metavar='num', type=int, default=8,
                   help='data loader workers per process (the reference hard-codes 2; ~6 feed one MI355X)')

    p = subparsers.add_parser('infer', help='run inference')
    p.add_argument('model', type=str, help='path to model')
    p.add_argument('--images', metavar='path', type=str, help='path to images', default='.')
    p.add_argument('--annotations', metavar='annotations', type=str, help='evaluate using provided annotations')
    p.add_argument('--output', metavar='file', type=str, nargs='+', help='save detections to specified JSON file(s)',
                   default=['detections.json'])
    p.add_argument('--batch', metavar='size', type=int, help='batch size', default=2 * devcount)
    p.add_argument('--resize', metavar='scale', type=int, help='resize to given size', default=800)
    p.add_argument('--max-size', metavar='max', type=int, help='maximum resizing size', default=1333)
    p.add_argument('--with-apex', help='(dropped dependency: refused)', action='store_true')
    p.add_argument('--with-dali', help='(dropped dependency: refused)', action='store_true')
    p.add_argument('--full-precision', help='inference in full precision', action='store_true')
    p.add_argument('--rotated-bbox', help='inference using a rotated bounding box model', action='store_true')
    p.add_argument('--workers', metavar='num', type=int, default=8,
                   help='data loader workers per process (the reference hard-codes 2; ~6 feed one MI355X)')

    p = subparsers.add_parser('export', help='export a model into a TensorRT engine (dropped: refused)')
    p.add_argument('model', type=str, help='path to model')
    p.add_argument('export', type=str, help='path to exported output')
    p.add_argument('--size', metavar='height width', type=int, nargs='+', default=[1280])
    p.add_argument('--full-precision', action='store_true')
    p.add_argument('--int8', action='store_true')
    p.add_argument('--calibration-batches', metavar='size', type=int, default=2)
    p.add_argument('--calibration-images', metavar='path', type=str, default='')
    p.add_argument('--calibration-table', metavar='path', type=str, default='')
    p.add_argument('--verbose', action='store_true')
    p.add_argument('--rotated-bbox', action='store_true')
    p.add_argument('--dynamic-batch-opts', metavar='value value value', type=int, nargs=3, default=[1, 8, 16])
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
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank)))
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
