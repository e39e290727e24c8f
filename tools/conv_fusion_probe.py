#!/usr/bin/env python
"""Probe (GPU): what does PyTorch-ROCm/MIOpen give for conv + bias + ReLU at the shapes of the
RetinaNet heads/backbone -- separate kernels vs torch.miopen_convolution_relu -- and does replaying
the whole forward from a hipGraph help?  Decides what odtk.model's inference fusion uses."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = 'cuda'

def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [('head3x3 P3', 8, 256, 256, 100, 160, 3, 1), ('head3x3 P4', 8, 256, 256, 50, 80, 3, 1),
          ('cls_out P3', 8, 256, 720, 100, 160, 3, 1), ('res 1x1 64->256', 8, 64, 256, 200, 320, 1, 1),
          ('res 3x3 128', 8, 128, 128, 100, 160, 3, 1), ('res 1x1 512->128', 8, 512, 128, 100, 160, 1, 1)]
for name, b, ci, co, h, w, k, s in shapes:
    x = torch.randn(b, ci, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, k, k, device=dev, dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(co, device=dev, dtype=torch.bfloat16)
    pad = k // 2
    flops = 2 * b * co * h * w * ci * k * k
    t_conv = timeit(lambda: F.conv2d(x, wt, None, s, pad))
    t_sep = timeit(lambda: F.relu(F.conv2d(x, wt, bias, s, pad)))
    try:
        t_fused = timeit(lambda: torch.miopen_convolution_relu(x, wt, bias, [s, s], [pad, pad], [1, 1], 1))
        ref = F.relu(F.conv2d(x, wt, bias, s, pad)); got = torch.miopen_convolution_relu(x, wt, bias, [s, s], [pad, pad], [1, 1], 1)
        err = (ref.float() - got.float()).abs().max().item()
    except Exception as e:
        t_fused, err = float('nan'), str(e)[:80]
    print('%-18s conv %.1f us (%.0f TF/s)  conv+bias+relu separate %.1f us  miopen fused %.1f us  maxerr %s' % (
        name, t_conv, flops / t_conv / 1e6, t_sep, t_fused, err))

# whole-model: eager vs hipGraph replay
from odtk.model import Model
torch.manual_seed(0)
m = Model('ResNet50FPN'); m.initialize(None)
m = m.to(dev).to(memory_format=torch.channels_last).eval()
x = torch.randn(8, 3, 800, 1280, device=dev).contiguous(memory_format=torch.channels_last)
def step():
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        return m(x)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print('eager: %.2f ms/step' % ((time.perf_counter() - t0) * 100))
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = step()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print('hipGraph replay: %.2f ms/step' % ((time.perf_counter() - t0) * 100))
except Exception as e:
    print('graph capture failed:', repr(e)[:300])
