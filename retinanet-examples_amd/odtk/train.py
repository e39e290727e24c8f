"""Data-parallel training of RetinaNet on MI355X: one process per GPU, torch DDP over RCCL/xGMI
(`backend="nccl"` IS RCCL on ROCm), gradient all-reduce bucketed and overlapped with backward.

Mirrors the behaviour of the reference's odtk/train.py (SGD momentum 0.9 / wd 1e-4 :33-34, LambdaLR
warm-up + milestones :52-57, autocast + GradScaler :91,107-121, frozen BN :29, channels_last :31,
divergence check :136-138, checkpoint every 60 s :145-183) without apex.  What is MI355X-specific:

  * DDP is built for a point-to-point xGMI fabric: buffers are constants (frozen BN) so
    `broadcast_buffers=False`; `gradient_as_bucket_view=True` (no grad<->bucket copies);
    `static_graph=True`; 151.7 MB of fp32 gradients per step for RN50FPN go out in ~25 MB buckets
    while backward is still running.
  * the two per-step loss all-reduces of the reference (:127-131) are ONE 3-element all-reduce,
    issued only on logging steps; which steps those are is decided from the iteration counter alone,
    so every rank enters the collective at the same iteration (see `train`).

The data source is any iterator of (images [B,3,H,W], targets [B,N,5|6] padded with -1) batches;
`SyntheticBatches` is the seeded stand-in used by tests and benchmarks (no dataset exists here).
"""
import math
import time

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel
from torch.optim import SGD
from torch.optim.lr_scheduler import LambdaLR

from .backbones.layers import convert_fixedbn_model
from .utils import ignore_sigint, post_metrics


class SyntheticBatches:
    """Seeded random images + boxes in the reference's target format (data.py:154-161: per image a
    [N, 5] table of x, y, w, h, class, padded with -1 rows), sharded by rank."""

    def __init__(self, batch, height, width, classes=80, max_boxes=20, seed=0, rank=0, world=1, device='cpu',
                 length=1 << 30, rotated=False):
        if batch % world:
            raise RuntimeError('Batch size should be a multiple of the number of GPUs')
        self.per_rank, self.h, self.w, self.classes, self.max_boxes = batch // world, height, width, classes, max_boxes
        self.gen = torch.Generator().manual_seed(seed * 1009 + rank)
        self.device, self.length, self.rotated = device, length, rotated   # rotated: [N, 6] rows x, y, w, h, theta, class

    def __len__(self):
        return self.length

    def __iter__(self):
        for _ in range(self.length):
            yield self.batch()

    def batch(self):
        g, b = self.gen, self.per_rank
        data = torch.randn(b, 3, self.h, self.w, generator=g)
        target = torch.full((b, self.max_boxes, 6 if self.rotated else 5), -1.0)
        for i in range(b):
            n = int(torch.randint(1, self.max_boxes + 1, (1,), generator=g))
            wh = torch.rand(n, 2, generator=g) * torch.tensor([min(400., self.w * .8) - 32, min(400., self.h * .8) - 32]) + 32
            xy = torch.rand(n, 2, generator=g) * (torch.tensor([float(self.w), float(self.h)]) - wh)
            cls = torch.randint(0, self.classes, (n, 1), generator=g).float()
            if self.rotated:
                theta = (torch.rand(n, 1, generator=g) - 0.5) * 1.0472        # +- 30 degrees
                target[i, :n] = torch.cat([xy, wh, theta, cls], 1)
            else:
                target[i, :n] = torch.cat([xy, wh, cls], 1)
        return data.to(self.device), target.to(self.device)


def lr_schedule(warmup, milestones, gamma):
    """reference train.py:52-56: linear warm-up from 0.1x, then gamma per passed milestone."""
    def schedule(step):
        if warmup and step <= warmup:
            return 0.9 * step / warmup + 0.1
        return gamma ** len([m for m in milestones if m <= step])
    return schedule


def prepare(model, device, lr=0.01, world=1, rank=0, warmup=1000, milestones=(), gamma=0.1, state=None,
            bucket_cap_mb=25, weight_decay=0.0001, frozen_bn=True):
    """Frozen-BN conversion, channels_last, optimizer, DDP wrapper, LR schedule (reference train.py:29-59).
    frozen_bn=False (not a reference option): the batch-norm layers stay live -- what training a backbone from its random
    initialisation needs; the reference never does that (it starts from ImageNet weights, resnet.py:20-22), this image has no
    weights to start from (tests/test_gpu_trained_ap.py pre-trains its detector this way, then fine-tunes frozen)."""
    if frozen_bn:
        model = convert_fixedbn_model(model)
    model = model.to(device)
    if device.type == 'cuda':
        model = model.to(memory_format=torch.channels_last)
    model.freeze_unused_params()
    # every parameter, frozen ones included, exactly like the reference (train.py:34): the param-group layout is
    # part of the checkpoint format (optimizer.load_state_dict), and SGD skips parameters without a gradient
    optimizer = SGD(model.parameters(), lr=lr, weight_decay=weight_decay, momentum=0.9)
    net = model
    if world > 1:
        net = DistributedDataParallel(model, device_ids=[device.index] if device.type == 'cuda' else None,
                                      broadcast_buffers=False, gradient_as_bucket_view=True, static_graph=True,
                                      bucket_cap_mb=bucket_cap_mb)
    model.train()
    if state and 'optimizer' in state:
        optimizer.load_state_dict(state['optimizer'])
    scheduler = LambdaLR(optimizer, lr_schedule(warmup, list(milestones), gamma))
    if state and 'scheduler' in state:
        scheduler.load_state_dict(state['scheduler'])
    return model, net, optimizer, scheduler


def train_step(net, optimizer, scheduler, scaler, data, target, amp_dtype=None):
    """One optimisation step; returns (cls_loss, box_loss) as detached tensors (no host sync)."""
    optimizer.zero_grad(set_to_none=True)
    use_amp = amp_dtype is not None and data.is_cuda
    with torch.autocast(data.device.type, dtype=amp_dtype, enabled=use_amp):
        cls_loss, box_loss = net([data, target])
    loss = cls_loss + box_loss
    if scaler is not None:
        scaler.scale(loss).backward()          # DDP: bucketed all-reduce overlaps with this backward
        scaler.step(optimizer)
        scaler.update()
    else:
        loss.backward()
        optimizer.step()
    scheduler.step()
    return cls_loss.detach(), box_loss.detach()


def reduce_losses(cls_loss, box_loss, world, extra=None):
    """Mean over ranks of both losses with ONE collective (reference: two per step, train.py:127-131).
    `extra`: an optional scalar that only rank 0 contributes (summed, so every rank reads rank 0's value)
    riding in the same message."""
    parts = [cls_loss.detach().float().reshape(()), box_loss.detach().float().reshape(())]
    if extra is not None:
        parts.append(torch.as_tensor(float(extra), dtype=torch.float32, device=parts[0].device))
    both = torch.stack(parts)
    if world > 1:
        dist.all_reduce(both)
        both[:2] /= world
    return both


def train_batches(model, state, batches, iterations, device, lr=0.01, warmup=1000, milestones=(), gamma=0.1, world=1,
                  rank=0, mixed_precision=True, log_every=60.0, save_path=None, verbose=True, log_interval=None,
                  weight_decay=0.0001, validate=None, val_iterations=None, on_report=None, frozen_bn=True):
    """The training loop of reference train.py:18-214 over any source of (images, targets) batches.

    `validate(net, iteration)` is called on EVERY rank (it gathers detections) after iteration `iterations` and after every
    `val_iterations`-th one (reference train.py:185); `on_report(iteration, focal, box, seconds_per_step, lr)` on
    rank 0 at each logging step (metrics / scalar logs).

    Logging / checkpoint cadence.  The reference all-reduces both losses and tests them on the host EVERY
    step (train.py:126-138: two collectives + one host sync per iteration).  Here the losses are summed on
    the device and reduced once per logging step.  Whether a step is a logging step must be the SAME
    decision on every rank (a collective that only some ranks enter would pair with another rank's
    gradient all-reduce), so it is a function of the iteration counter only: every `interval` iterations,
    where `interval` starts at `log_interval` (default 10) and is re-derived at each logging step from
    RANK 0's measured step time (target: one report per `log_every` seconds) and handed to every rank
    inside the loss all-reduce itself."""
    model, net, optimizer, scheduler = prepare(model, device, lr, world, rank, warmup, milestones, gamma, state,
                                              weight_decay=weight_decay, frozen_bn=frozen_bn)
    amp_dtype = torch.float16 if (mixed_precision and device.type == 'cuda') else None
    scaler = torch.amp.GradScaler('cuda', enabled=amp_dtype is not None) if amp_dtype is not None else None
    iteration = state.get('iteration', 0) if state else 0
    interval = max(1, int(log_interval)) if log_interval else 10
    adaptive = log_interval is None
    next_log = iteration + interval
    t_mark, n_mark = time.time(), 0
    cls_sum = box_sum = None

    def report():
        nonlocal interval, next_log, t_mark, n_mark, cls_sum, box_sum
        now = time.time()
        per_step = (now - t_mark) / max(n_mark, 1)
        proposal = max(1, min(1000, int(round(log_every / max(per_step, 1e-6))))) if adaptive else interval   # the divergence test
        # below runs once per interval (the reference: every step, train.py:132-138): never let it drift past 1000 steps
        both = reduce_losses(cls_sum / n_mark, box_sum / n_mark, world, proposal if rank == 0 else 0.0)
        cls_mean, box_mean, agreed = (float(v) for v in both)       # the only host sync of the interval
        if not math.isfinite(cls_mean + box_mean):
            raise RuntimeError('Loss is diverging!\nTry lowering the learning rate.')
        if rank == 0 and verbose:
            print('[{:{w}}/{}] focal loss: {:.3f}, box loss: {:.3f}, {:.3f}s/{}-batch, {:.1f} im/s, lr: {:.2g}'.format(
                iteration, iterations, cls_mean, box_mean, per_step, int(round(seen / max(n_mark, 1))), seen / max(now - t_mark, 1e-9),
                scheduler.get_last_lr()[0], w=len(str(iterations))), flush=True)
        if rank == 0 and on_report is not None:
            on_report(iteration, cls_mean, box_mean, per_step, scheduler.get_last_lr()[0])
        if rank == 0 and save_path:
            with ignore_sigint():
                model.save({'path': save_path, 'iteration': iteration, 'optimizer': optimizer.state_dict(),
                            'scheduler': scheduler.state_dict()})
        interval = max(1, int(round(agreed)))
        next_log = iteration + interval
        t_mark, n_mark, cls_sum, box_sum = time.time(), 0, None, None

    seen = 0
    while iteration < iterations:                                    # epochs (reference train.py:93)
        progressed = False
        for data, target in batches:
            if iteration >= iterations:
                break
            progressed = True
            if device.type == 'cuda':
                data = data.contiguous(memory_format=torch.channels_last)
            cls_loss, box_loss = train_step(net, optimizer, scheduler, scaler, data.to(device), target.to(device), amp_dtype)
            iteration += 1
            n_mark += 1
            seen += data.shape[0] * world
            cls_sum = cls_loss if cls_sum is None else cls_sum + cls_loss
            box_sum = box_loss if box_sum is None else box_sum + box_loss
            if iteration >= next_log or iteration == iterations:
                report()
                seen = 0
            if validate is not None and (iteration == iterations or (val_iterations and iteration % val_iterations == 0)):
                validate(net, iteration)
                net.train()
                t_mark = time.time() if n_mark == 0 else t_mark     # validation time is not step time
        if not progressed:                                           # empty data source: nothing will ever change
            break
    if n_mark:                                                       # the source ran dry between two reports
        report()
    return iteration


class ScalarLog:
    """`logdir` of the reference is a TensorBoard directory (train.py:81-86); tensorboard is not installed here,
    so scalars go through `SummaryWriter` when it imports and into `<logdir>/scalars.jsonl` otherwise."""

    def __init__(self, logdir):
        import json
        import os
        os.makedirs(logdir, exist_ok=True)
        self._json, self.writer, self.file = json, None, None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer = SummaryWriter(log_dir=logdir)
        except Exception:                                           # noqa: BLE001 -- any import failure
            self.file = open(os.path.join(logdir, 'scalars.jsonl'), 'a')

    def add_scalar(self, tag, value, iteration):
        if self.writer is not None:
            self.writer.add_scalar(tag, value, iteration)
        else:
            self.file.write(self._json.dumps({'tag': tag, 'value': float(value), 'iteration': int(iteration)}) + '\n')
            self.file.flush()

    def close(self):
        (self.writer or self.file).close()


VALIDATION_TAGS = ['Validation_Precision/mAP', 'Validation_Precision/mAP@0.50IoU', 'Validation_Precision/mAP@0.75IoU',
                   'Validation_Precision/mAP (small)', 'Validation_Precision/mAP (medium)', 'Validation_Precision/mAP (large)',
                   'Validation_Recall/mAR (max 1 Dets)', 'Validation_Recall/mAR (max 10 Dets)',
                   'Validation_Recall/mAR (max 100 Dets)', 'Validation_Recall/mAR (small)', 'Validation_Recall/mAR (medium)',
                   'Validation_Recall/mAR (large)']


def train(model, state, path, annotations, val_path, val_annotations, resize, max_size, jitter, batch_size, iterations,
          val_iterations, lr, warmup, milestones, gamma, rank=0, world=1, mixed_precision=True, with_apex=False,
          use_dali=False, verbose=True, metrics_url=None, logdir=None, rotate_augment=False, augment_brightness=0.0,
          augment_contrast=0.0, augment_hue=0.0, augment_saturation=0.0, regularization_l2=0.0001, rotated_bbox=False,
          absolute_angle=False, num_workers=2, device=None):
    """Train `model` on the images under `path` -- the reference's `train.train` (train.py:18-214), same
    arguments: COCO-style annotations through odtk/data.py (`jitter` = the range of short-side sizes),
    `train_batches` above, periodic validation through `infer.infer`, checkpoints to `state['path']`, scalars
    to `logdir`, metrics to `metrics_url`.  `with_apex` / `use_dali` are errors (dropped dependencies)."""
    from . import infer as infer_module
    from .data import DataIterator, RotatedDataIterator
    if use_dali or with_apex:
        raise RuntimeError('DALI and apex are not part of this build (use the default loader / torch autocast)')
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    if verbose:
        print('Preparing dataset...')
    extra = {'absolute_angle': absolute_angle} if rotated_bbox else {}
    data_iterator = (RotatedDataIterator if rotated_bbox else DataIterator)(
        path, jitter, max_size, batch_size, model.stride, world, annotations, training=True,
        rotate_augment=rotate_augment, augment_brightness=augment_brightness, augment_contrast=augment_contrast,
        augment_hue=augment_hue, augment_saturation=augment_saturation, device=device, num_workers=num_workers, **extra)
    if verbose:
        print(data_iterator)
        print('    device: {} {}'.format(world, 'cpu' if device.type == 'cpu' else 'GPU' if world == 1 else 'GPUs'))
        print('     batch: {}, precision: {}'.format(batch_size, 'mixed' if mixed_precision else 'full'))
        print(' BBOX type:', 'rotated' if rotated_bbox else 'axis aligned')
        print('Training model for {} iterations...'.format(iterations))
    is_master = rank == 0
    log = ScalarLog(logdir) if (is_master and logdir is not None) else None
    if log is not None and verbose:
        print('Writing logs to: {}'.format(logdir))
    def on_report(iteration, focal, box, per_step, learning_rate):
        if log is not None:
            log.add_scalar('focal_loss', focal, iteration)
            log.add_scalar('box_loss', box, iteration)
            log.add_scalar('learning_rate', learning_rate, iteration)
        if metrics_url:
            post_metrics(metrics_url, {'focal loss': focal, 'box loss': box, 'im_s': batch_size / max(per_step, 1e-9),
                                       'lr': learning_rate})

    def validate(net, iteration):
        stats = infer_module.infer(net, val_path, None, resize, max_size, batch_size, annotations=val_annotations,
                                   mixed_precision=mixed_precision, is_master=is_master, world=world,
                                   is_validation=True, verbose=False, rotated_bbox=rotated_bbox, num_workers=num_workers)
        validate.last = stats
        if log is not None and stats is not None and not isinstance(stats, int):
            for tag, value in zip(VALIDATION_TAGS, stats):
                log.add_scalar(tag, value, iteration)

    validate.last = None
    try:
        done = train_batches(model, state or {}, data_iterator, iterations, device, lr=lr, warmup=warmup,
                             milestones=milestones, gamma=gamma, world=world, rank=rank, mixed_precision=mixed_precision,
                             save_path=(state or {}).get('path'), verbose=verbose, weight_decay=regularization_l2,
                             validate=validate if val_annotations else None, val_iterations=val_iterations,
                             on_report=on_report)
    finally:
        if log is not None:
            log.close()
    return done, validate.last
