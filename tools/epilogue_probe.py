#!/usr/bin/env python
"""bias_act_ (in-place bias + ReLU epilogue, csrc/epilogue.hpp) on the activation sizes of the RN50FPN bs-8 step, one size at a
time, rotating through three buffers (nothing served from the Infinity Cache): dispatch-timestamp time and GB/s per size."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
from odtk import _C

shapes = [(8, 64, 200, 320), (8, 128, 100, 160), (8, 256, 100, 160), (8, 256, 50, 80), (8, 512, 25, 40), (8, 256, 25, 40),
          (8, 256, 13, 20), (8, 256, 7, 10), (8, 36, 100, 160)]
for res in (False, True):
    for shape in shapes:
        n = 1
        for s in shape:
            n *= s
        copies = max(3, int(600e6 // (n * 2)))            # > 256 MiB in rotation
        ys = [torch.randn(shape, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(min(copies, 12))]
        r = torch.randn(shape, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last) if res else None
        bias = torch.randn(shape[1], device='cuda')
        for y in ys:
            _C.bias_act_(y, bias, r, True)
        torch.cuda.synchronize()
        _C.profile_enable(True, ('bias_act_kernel',))
        _C.profile_collect()
        for it in range(24):
            _C.bias_act_(ys[it % len(ys)], bias, r, True)
        torch.cuda.synchronize()
        _C.profile_enable(False)
        ms, k = _C.profile_collect()['bias_act_kernel']
        us = ms / k * 1e3
        nbytes = n * 2 * (3 if res else 2)
        print('%-22s residual=%d  %7.2f MB  %7.2f us  %7.1f GB/s' % ('x'.join(map(str, shape)), res, nbytes / 1e6, us, nbytes / us / 1e3), flush=True)
