#!/usr/bin/env python
"""1-GPU training-step probe (BASELINE config 3 shape per GPU: RN50FPN, 800x1280, 2 images/GPU):
step time, and the share of target assignment (snap_to_anchors) in it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
torch.backends.cudnn.benchmark = True
from odtk import train as T, box
from odtk.model import Model

amp = '--amp' in sys.argv
dev = torch.device('cuda')
torch.manual_seed(0)
m = Model('ResNet50FPN'); m.initialize(None)
model, net, opt, sched = T.prepare(m, dev, lr=0.001, world=1)
scaler = torch.amp.GradScaler('cuda') if amp else None
batches = T.SyntheticBatches(2, 800, 1280, seed=1, device='cuda')
data, target = batches.batch()
data = data.contiguous(memory_format=torch.channels_last)
for _ in range(4):
    T.train_step(net, opt, sched, scaler, data, target, torch.float16 if amp else None)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 10
for _ in range(n):
    c, b = T.train_step(net, opt, sched, scaler, data, target, torch.float16 if amp else None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('train step (%s): %.1f ms/step = %.1f img/s, losses %.3f %.3f' % ('amp fp16' if amp else 'fp32', dt * 1e3, 2 / dt, float(c), float(b)))
# target assignment alone
sizes = [(100, 160, 8), (50, 80, 16), (25, 40, 32), (13, 20, 64), (7, 10, 128)]
def assign():
    for h, w, s in sizes:
        anchors = model.level_anchors(s).cuda()
        for t in target:
            box.snap_to_anchors(t[t[:, -1] > -1], [w * s, h * s], s, anchors, 80, dev, [0.4, 0.5])
for _ in range(3): assign()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): assign()
torch.cuda.synchronize()
print('target assignment (torch ops, 5 levels x 2 images): %.2f ms per step' % ((time.perf_counter() - t0) * 100))
