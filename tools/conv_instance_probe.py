#!/usr/bin/env python
"""Every instance of libodtk_conv.so's list on ONE problem: time and name, fastest first (ODTK_CONV_INSTANCE forces an instance).
Next to it: the MIOpen convolution alone and MIOpen + odtk_bias_act on the same tensors.

    python tools/conv_instance_probe.py [--shape 8 256 100 160 256 3 1 1] [--top 12]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
import torch.nn.functional as F
from odtk import _C

ap = argparse.ArgumentParser()
ap.add_argument('--shape', type=int, nargs=8, default=[8, 256, 100, 160, 256, 3, 1, 1], help='batch c_in h w c_out k stride pad')
ap.add_argument('--top', type=int, default=12)
ap.add_argument('--dtype', default='bf16')
a = ap.parse_args()
b, c, h, w, k, ks, stride, pad = a.shape
dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16}[a.dtype]
torch.backends.cudnn.benchmark = True
g = torch.Generator().manual_seed(0)
x = (torch.randn(b, c, h, w, generator=g) * 0.5).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
wt = (torch.randn(k, c, ks, ks, generator=g) * 0.05).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
bias = torch.randn(k, generator=g).to(dtype).cuda()
bias32 = bias.float()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for _ in range(3):
    F.conv2d(x, wt, None, stride, pad)
t_conv = timed(lambda: F.conv2d(x, wt, None, stride, pad))
t_two = timed(lambda: _C.bias_act_(F.conv2d(x, wt, None, stride, pad), bias32, None, True))
print('problem %s %s: MIOpen convolution alone %.1f us, + odtk_bias_act %.1f us' % (a.shape, a.dtype, t_conv, t_two))
n = _C.conv_library().odtk_conv_instance_count(_C._DTYPES[dtype])
rows = []
for i in range(n):
    os.environ['ODTK_CONV_INSTANCE'] = str(i)
    try:
        t = timed(lambda: _C.conv_bias_act(x, wt, bias, stride, pad, True), reps=5)
    except RuntimeError:
        continue
    rows.append((t, i, _C.conv_last_plan().split(' ', 3)[-1]))
os.environ.pop('ODTK_CONV_INSTANCE', None)
rows.sort()
print('%d of %d instances take the problem; fastest first:' % (len(rows), n))
for t, i, name in rows[:a.top]:
    print('%8.1f us  #%-3d %s' % (t, i, name))
