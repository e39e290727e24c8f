#!/usr/bin/env python
"""The loss kernels at their built-in launch shapes, 30 forward + 30 backward launches per dtype over three rotating input
sets at the training step's sizes -- meant to run under `rocprofv3 --kernel-trace --stats` (profiles/r03_loss_kernel_stats.*),
whose per-kernel averages must agree with tools/loss_probe.py's event-timed figures."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd'), os.path.join(ROOT, 'tools')]
import torch
from odtk import _C
from loss_probe import make_set, SIZES

for dtype in (torch.float32, torch.float16):
    sets = [make_set(dtype, 10 + i) for i in range(3)]
    gc = torch.full((len(SIZES),), 0.37, device='cuda')
    gb = torch.full((len(SIZES),), -1.9, device='cuda')
    for i in range(33):
        s = sets[i % 3]
        _C.retina_loss_levels_forward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11)
    for i in range(33):
        s = sets[i % 3]
        _C.retina_loss_levels_backward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11, gc, gb)
    torch.cuda.synchronize()
print('done')
