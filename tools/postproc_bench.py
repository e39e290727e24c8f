#!/usr/bin/env python
"""Post-processing-only micro-benchmark: decode x5 + NMS on device-resident synthetic head tensors
(no backbone), per-kernel times from the library's hipEvent hooks.  Used for kernel work and as the
small target for `rocprofv3 --pmc` passes.

    python tools/postproc_bench.py --kind sparse --batch 8 --iters 20
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from odtk import _C, box, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kind', default='sparse', choices=['sparse', 'dense', 'empty', 'saturated'])
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--sigma', type=float, default=None, help='std of the logits (overrides --kind; 0.6225 -> 0.4 %% pass 0.05)')
    ap.add_argument('--height', type=int, default=800)
    ap.add_argument('--width', type=int, default=1280)
    ap.add_argument('--anchors', type=int, default=9)
    ap.add_argument('--classes', type=int, default=80)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--ndet', type=int, default=100)
    ap.add_argument('--rotated', action='store_true')
    ap.add_argument('--threshold', type=float, default=0.05)
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16', 'fp16'])
    ap.add_argument('--logits', action='store_true', help='feed raw logits (sigmoid fused into the prefilter)')
    ap.add_argument('--channels-last', action='store_true')
    ap.add_argument('--bias', action='store_true', help='pass the last-conv biases to the kernels (cls_bias / box_bias)')
    ap.add_argument('--table', action='store_true', help='with --bias: also the precomputed threshold table (cls_thresholds)')
    ap.add_argument('--torch-baselines', action='store_true', help='also time torch read-only / copy passes')
    args = ap.parse_args()

    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(1234)
    nb = 6 if args.rotated else 4
    strides = (8, 16, 32, 64, 128)
    cls, dl = [], []
    for (h, w) in synthetic.level_shapes(args.height, args.width, strides):
        shape = (args.batch, args.anchors * args.classes, h, w)
        if args.kind == 'empty':
            c = torch.full(shape, 0.01, device=dev)
        elif args.kind == 'saturated':
            c = torch.ones(shape, device=dev)
        else:
            c = torch.randn(shape, generator=g, device=dev) * (args.sigma or synthetic.SIGMA[args.kind]) + synthetic.LOGIT_PRIOR
            if not args.logits:
                c = c.sigmoid()
        d = torch.randn((args.batch, args.anchors * nb, h, w), generator=g, device=dev) * 0.2
        tdt = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}[args.dtype]
        c, d = c.to(tdt), d.to(tdt)
        if args.channels_last:
            c = c.contiguous(memory_format=torch.channels_last)
            d = d.contiguous(memory_format=torch.channels_last)
        cls.append(c)
        dl.append(d)
    ratios, scales = [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)]
    if args.rotated:
        ang = [-math.pi / 6, 0, math.pi / 6]
        anchors = {s: box.generate_anchors_rotated(s, ratios, scales, ang) for s in strides}
    else:
        anchors = {s: box.generate_anchors(s, ratios, scales) for s in strides}
    if args.rotated:
        assert args.anchors == 27
    cand = [int(((c.float().sigmoid() if args.logits else c.float()) >= args.threshold).sum().item()) // args.batch
            for c in cls]

    cls_bias = box_bias = None
    if args.bias:
        cls_bias = torch.randn(args.anchors * args.classes, generator=g, device=dev) * 0.05
        box_bias = torch.randn(args.anchors * nb, generator=g, device=dev) * 0.05
    table = _C.prefilter_thresholds(cls_bias, tdt, args.threshold) if args.bias and args.table else None

    def run():
        return box.detect(cls, dl, list(strides), anchors, args.threshold, 1000, 0.5, args.ndet, args.rotated,
                          logits=args.logits, cls_bias=cls_bias, box_bias=box_bias, cls_thresholds=table)

    for _ in range(args.warmup):
        out = run()
    torch.cuda.synchronize()
    _C.profile_enable(True)
    _C.profile_collect()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        out = run()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.iters
    _C.profile_enable(False)
    prof = _C.profile_collect()
    n_scores = sum(c.numel() for c in cls)
    esize = cls[0].element_size()
    res = {'kind': args.kind, 'dtype': args.dtype, 'logits': args.logits, 'channels_last': args.channels_last,
           'threshold': args.threshold, 'batch': args.batch, 'candidates_per_image_per_level': cand,
           'detections': int((out[0] > 0).sum().item()), 'wall_us_per_call': round(wall * 1e6, 1),
           'kernels_us': {k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items() if v[1]},
           'kernels_us_per_call': {k: round(v[0] / args.iters * 1e3, 2) for k, v in prof.items() if v[1]},
           'select_us_per_call': round(sum(v[0] for k, v in prof.items() if k.startswith('select_')) / args.iters * 1e3, 2),
           'scores': n_scores, 'alg_bytes': esize * n_scores}
    t = prof['prefilter_scan_kernel']
    if t[1]:
        res['prefilter_GBps'] = round(esize * n_scores / (t[0] / t[1] * 1e-3) / 1e9, 1)
    if args.torch_baselines:
        def timeit(fn, n=10):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        big = cls[0]
        dst = torch.empty_like(big)
        res['torch_amax_P3_GBps'] = round(big.numel() * esize / timeit(lambda: big.amax()) / 1e3, 1)
        res['torch_copy_P3_GBps_rw'] = round(2 * big.numel() * esize / timeit(lambda: dst.copy_(big)) / 1e3, 1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
