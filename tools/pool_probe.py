#!/usr/bin/env python
"""Stem epilogue (bias + ReLU + 3x3/s2 max-pool, csrc/epilogue.hpp) at the bench's size: bs 8, 64 channels, 400x640 bf16.
Event-timed back to back; the full-size result is compared with bias_act_ + torch's max_pool2d."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
import torch.nn.functional as F
from odtk import _C

g = torch.Generator(device='cuda').manual_seed(0)
y = torch.randn(8, 64, 400, 640, device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
bias = torch.randn(64, device='cuda', generator=g)
got = _C.bias_act_maxpool(y, bias, True)
ref = F.max_pool2d(_C.bias_act_(y.clone(memory_format=torch.channels_last), bias, None, True), 3, 2, 1)
assert torch.equal(got, ref)
for _ in range(3):
    _C.bias_act_maxpool(y, bias, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 30
e0.record()
for _ in range(n):
    _C.bias_act_maxpool(y, bias, True)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
nbytes = y.numel() * 2 + got.numel() * 2
print('bias_act_maxpool bs8 64ch 400x640 bf16: %.1f us per call, %.0f GB/s over read-once + write (%d MB); == epilogue then pool' %
      (us, nbytes / us / 1e3, nbytes >> 20))
