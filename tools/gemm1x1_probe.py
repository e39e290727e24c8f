#!/usr/bin/env python
"""1x1 convolutions of the ResNet-50 bottlenecks at the headline shape (bs 8, 800x1280): MIOpen conv +
the HIP epilogue versus a hipBLASLt GEMM on the channels_last activation viewed as [N*H*W, C] with the
bias / ReLU in the GEMM epilogue.  Decides whether routing 1x1 convs through GEMM is worth it.

    python tools/gemm1x1_probe.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from odtk import _C  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    torch.backends.cudnn.benchmark = True
    dev = torch.device('cuda')
    b = 8
    # (H, W, Cin, Cout, count per forward, kind)
    shapes = [(200, 320, 64, 64, 1, 'conv1'), (200, 320, 256, 64, 2, 'conv1'), (200, 320, 64, 256, 3, 'conv3'),
              (100, 160, 512, 128, 3, 'conv1'), (100, 160, 128, 512, 4, 'conv3'),
              (50, 80, 1024, 256, 5, 'conv1'), (50, 80, 256, 1024, 6, 'conv3'),
              (25, 40, 2048, 512, 2, 'conv1'), (25, 40, 512, 2048, 3, 'conv3')]
    rows = []
    tot = {'conv+ep': 0.0, 'gemm_fused': 0.0}
    for (h, w, ci, co, cnt, kind) in shapes:
        x = torch.randn(b, ci, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(co, ci, 1, 1, device=dev) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
        bias = torch.randn(co, device=dev)
        bias16 = bias.bfloat16()
        res = torch.randn(b, co, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        m = b * h * w
        x2 = x.permute(0, 2, 3, 1).reshape(m, ci)
        assert x2.data_ptr() == x.data_ptr()
        w2t = wt.view(co, ci).t()
        res2 = res.permute(0, 2, 3, 1).reshape(m, co)
        r = {'shape': [m, ci, co], 'kind': kind, 'count': cnt}
        r['conv'] = timeit(lambda: F.conv2d(x, wt))
        if kind == 'conv1':
            r['conv+ep'] = timeit(lambda: _C.bias_act_(F.conv2d(x, wt), bias, None, True))
            r['gemm_fused'] = timeit(lambda: torch._addmm_activation(bias16, x2, w2t))
        else:
            r['conv+ep'] = timeit(lambda: _C.bias_act_(F.conv2d(x, wt), bias, res, True))
            r['gemm_fused'] = timeit(lambda: _C.bias_act_(torch.addmm(res2, x2, w2t).view(b, h, w, co).permute(0, 3, 1, 2),
                                                           bias, None, True))
            r['addmm_res'] = timeit(lambda: torch.addmm(res2, x2, w2t))
        r['mm'] = timeit(lambda: torch.mm(x2, w2t))
        # numerics of the GEMM route against the conv route
        ya = _C.bias_act_(F.conv2d(x, wt), bias, None, True).float()
        yb = torch._addmm_activation(bias16, x2, w2t).view(b, h, w, co).permute(0, 3, 1, 2).float()
        r['max_abs_diff'] = float((ya - yb).abs().max())
        for k in tot:
            tot[k] += r[k] * cnt
        rows.append({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
        print(rows[-1], flush=True)
    print(json.dumps({'per_forward_us': {k: round(v, 1) for k, v in tot.items()}}))


if __name__ == '__main__':
    main()
