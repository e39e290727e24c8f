#!/usr/bin/env python
"""Rotated training, settled: does the loss go non-finite because of the reference's initialisation
(model.py:121-122 puts the -4.6 class prior on all six box outputs: box loss starts at ~28) or because of a kernel?

Two replicas of the same seeded model take the same SGD steps on the same batches: one through the fused path (one HIP
target launch + the fused focal / smooth-L1 kernel), one through the reference-style path (per-image
snap_to_anchors_rotated + torch losses).  Printed per step: both losses of both replicas.  Then the same again with the box
head's bias at (0, 0, 0, 0, sin = 0, cos = 1) -- what `bench.py --unit-rotation` uses -- instead of the prior.

    python tools/rotated_train_probe.py [--steps 60] [--lr 0.001] [--backbone ResNet18FPN]
"""
import argparse
import copy
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
from odtk import train as T
from odtk.model import Model


def unit_rotation_bias(model):
    with torch.no_grad():
        bias = model.box_head[-1].bias.view(model.num_anchors, 6)
        bias.zero_()
        bias[:, 5] = 1.0


def trajectory(backbone, steps, lr, unit, height=384, width=512, classes=20, seed=0, momentum=0.9, bench_like=False):
    torch.manual_seed(seed)
    base = Model(backbone, classes=classes, rotated_bbox=True)
    base.initialize(None)
    if unit:
        unit_rotation_bias(base)
    src = T.SyntheticBatches(2, height, width, classes=classes, max_boxes=20 if bench_like else 12, seed=0 if bench_like else 3,
                             device='cuda', rotated=True)
    batches = [src.batch() for _ in range(4)]
    rows = {}
    for fused in (True, False):
        if bench_like:
            # exactly bench.py's leg: frozen BN, lr 0.01 behind the reference's 1000-step warm-up (x 0.1 at step 0), fp32
            m, net, opt, sched = T.prepare(copy.deepcopy(base), torch.device('cuda'), lr=0.01, world=1, rank=0, warmup=1000)
        else:
            m = copy.deepcopy(base).cuda().to(memory_format=torch.channels_last).train()
            opt = torch.optim.SGD(m.parameters(), lr=lr, momentum=momentum, weight_decay=1e-4)
            sched = None
        m.fused_loss = fused
        out = []
        for step in range(steps):
            d, t = batches[step % len(batches)]
            if bench_like:
                c, b = T.train_step(m, opt, sched, None, d.contiguous(memory_format=torch.channels_last), t, None)
            else:
                opt.zero_grad(set_to_none=True)
                c, b = m([d.contiguous(memory_format=torch.channels_last), t])
                (c + b).backward()
                opt.step()
            out.append((float(c.detach()), float(b.detach())))
            if not math.isfinite(out[-1][0] + out[-1][1]):
                break
        rows['fused' if fused else 'torch'] = out
        del m, opt
        torch.cuda.empty_cache()
    return rows


def first_nonfinite(rows):
    return next((i for i, (c, b) in enumerate(rows) if not math.isfinite(c + b)), None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--lr', type=float, default=0.001)      # what bench.py's leg runs at (0.01 x the 0.1 warm-up factor)
    ap.add_argument('--backbone', default='ResNet18FPN')
    ap.add_argument('--height', type=int, default=384)
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--classes', type=int, default=20)
    ap.add_argument('--bench-like', action='store_true', help="bench.py's rotated training leg: ResNet50FPN, 800x1280, 80 classes")
    a = ap.parse_args()
    if a.bench_like:
        a.backbone, a.height, a.width, a.classes = 'ResNet50FPN', 800, 1280, 80
    for unit in (False, True):
        rows = trajectory(a.backbone, a.steps, a.lr, unit, a.height, a.width, a.classes, bench_like=a.bench_like)
        print('== %s %dx%d, %d classes, lr %g, box-head bias: %s ==' % (a.backbone, a.height, a.width, a.classes, a.lr, 'unit rotation (0,0,0,0,0,1)' if unit else "reference prior -4.6 (model.py:121-122)"))
        print('step   fused: focal      box   |   torch: focal      box')
        for i in range(max(len(rows['fused']), len(rows['torch']))):
            f = rows['fused'][i] if i < len(rows['fused']) else (float('nan'),) * 2
            t = rows['torch'][i] if i < len(rows['torch']) else (float('nan'),) * 2
            print('%4d   %12.5f %9.4f   |   %12.5f %9.4f' % (i, f[0], f[1], t[0], t[1]))
        print('first non-finite step: fused %s, torch %s' % (first_nonfinite(rows['fused']), first_nonfinite(rows['torch'])), flush=True)


if __name__ == '__main__':
    main()
