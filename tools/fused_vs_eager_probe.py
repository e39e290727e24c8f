"""Numerical fidelity of the fused bf16 engine vs the eager autocast graph vs fp32,
and the rounding mode of the bf16 convolution kernels.

    python tools/fused_vs_eager_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd')):
    sys.path.insert(0, p)
import bench
from odtk.model import Model
from odtk.fused import FusedRetinaNet
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda')
torch.manual_seed(0)
model = Model('ResNet50FPN', classes=80); model.initialize(None)
model = model.to(dev).to(memory_format=torch.channels_last).eval()
x = torch.randn(8, 3, 800, 1280, device=dev).contiguous(memory_format=torch.channels_last)
s0 = bench.calibrate_cls_head(model, x, 0.573, torch.bfloat16)
eng = FusedRetinaNet(model, dtype=torch.bfloat16).to(dev)
with torch.no_grad():
    with torch.autocast('cuda', dtype=torch.bfloat16):
        c_ref, b_ref = model.heads(x)
    c_f, b_f = eng.heads(x)
for l, (r, f) in enumerate(zip(c_ref, c_f)):
    r, f = r.float(), f.float()
    print('P%d ref mean %.3f std %.3f cand/img %d | fused mean %.3f std %.3f cand/img %d | cos %.5f maxdiff %.3f' % (
        l + 3, r.mean(), r.std(), (r.sigmoid() >= 0.05).sum() // 8, f.mean(), f.std(), (f.sigmoid() >= 0.05).sum() // 8,
        torch.nn.functional.cosine_similarity((r - r.mean()).flatten(), (f - f.mean()).flatten(), dim=0), (r - f).abs().max()))
# fp32 truth and the rounding mode of the bf16 convolution kernels
with torch.no_grad():
    c32, _ = model.heads(x)            # no autocast: fp32 convolutions
for l, (t, r, f) in enumerate(zip(c32, c_ref, c_f)):
    print('P%d fp32 std %.3f | autocast/fp32 %.4f | fused/fp32 %.4f' % (l + 3, t.std(), r.float().std() / t.std(), f.float().std() / t.std()))
import torch.nn.functional as F
xa = torch.randn(4, 64, 64, 64, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
for k in (1, 3):
    w = (torch.randn(64, 64, k, k, device=dev) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    y16 = F.conv2d(xa, w, padding=k // 2).float()
    y32 = F.conv2d(xa.float(), w.float(), padding=k // 2)
    rne = y32.bfloat16().float()
    trunc = (y32.view(torch.int32) & -65536).view(torch.float32)
    print('conv %dx%d bf16: equals RNE(fp32) on %.1f %%, equals truncation on %.1f %%, mean(|y16| - |y32|)/mean|y32| = %.2e' % (
        k, k, 100 * (y16 == rne).float().mean(), 100 * (y16 == trunc).float().mean(), ((y16.abs() - y32.abs()).mean() / y32.abs().mean())))
