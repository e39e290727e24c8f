#!/usr/bin/env python
"""Rotated training: the fused path (one target launch + fused loss) against the reference-style path (per-image
snap_to_anchors_rotated + torch losses) on the same model and batches; loss trajectory of a few SGD steps."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
from odtk import train as T
from odtk.model import Model

torch.manual_seed(0)
m = Model('ResNet18FPN', classes=20, rotated_bbox=True)
m.initialize(None)
m = m.cuda().to(memory_format=torch.channels_last).train()
src = T.SyntheticBatches(2, 384, 512, classes=20, max_boxes=12, seed=3, device='cuda', rotated=True)
data, target = src.batch()
data = data.contiguous(memory_format=torch.channels_last)
for fused in (True, False):
    m.fused_loss = fused
    m.zero_grad(set_to_none=True)
    c, b = m([data, target])
    (c + b).backward()
    gn = sum(float(p.grad.float().pow(2).sum()) for p in m.parameters() if p.grad is not None) ** 0.5
    print('fused' if fused else 'torch', 'cls %.6f box %.6f grad-norm %.6f' % (float(c), float(b), gn), flush=True)
m.fused_loss = True
opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
for step in range(12):
    d, t = src.batch()
    opt.zero_grad(set_to_none=True)
    c, b = m([d.contiguous(memory_format=torch.channels_last), t])
    (c + b).backward()
    opt.step()
    print('step %d cls %.5f box %.5f' % (step, float(c), float(b)), flush=True)
