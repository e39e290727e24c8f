#!/usr/bin/env python
"""Long differential fuzz of the FUSED fast path -- what the timed engine hands the post-processing: raw 16-bit logits in
channels_last, the last convolutions' biases folded into the kernels, the prefilter's precomputed per-channel threshold table --
against the STRICT op (fp32 NCHW scores) fed with what the reference pipeline materialises first, by torch on the GPU:
scores = sigmoid(float(raw) + bias) rounded to the head dtype, deltas = float(raw) + bias.  Bit for bit, indices included
(tests/test_gpu_fused.py::test_head_bias_folded_into_the_kernels at a few shapes; here: random anchor / class counts with
A x C % 8 == 0, 1..4 levels of 1 x 1 .. 150 x 200 cells, batch 1..4, bf16 / fp16, thresholds 0.01..0.9, top_n 1..2000, bias
spreads, with and without biases / table, axis-aligned and rotated).  Both sides run on the GPU: thousands of cases in minutes.
The strict op itself is pinned to the oracle by tools/decode_fuzz_long.py and the parity suites.

    python tools/fused_fuzz_long.py --seeds 0:3000
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def case(seed):
    r = np.random.default_rng(12000 + seed)
    rotated = r.random() < 0.2
    if rotated:
        a, c = 27, int(r.choice([8, 16]))
    else:
        a, c = [(9, 8), (9, 16), (9, 80), (3, 8), (1, 8), (2, 4), (4, 2), (8, 1), (6, 4)][int(r.integers(0, 9))]
    levels = int(r.integers(1, 5))
    b = int(r.integers(1, 5))
    big = r.random() < 0.35
    shapes = [(int(r.integers(1, 150 if big else 30)), int(r.integers(1, 200 if big else 30))) for _ in range(levels)]
    strides = [int(r.choice([s for s in (4, 8, 16, 32) if s * max(h, w) <= 3200] or [4])) for (h, w) in shapes]
    thr = float(r.choice([0.01, 0.05, 0.05, 0.3, 0.5, 0.9]))
    top_n = int(r.choice([1, 64, 300, 1000, 1000, 2000]))
    dtype = torch.bfloat16 if r.random() < 0.6 else torch.float16
    spread = float(r.choice([0.6, 1.2, 3.0]))
    bias_mode = str(r.choice(['both', 'both', 'cls', 'none']))
    table = bool(r.integers(0, 2))
    return rotated, a, c, b, shapes, strides, thr, top_n, dtype, spread, bias_mode, table


def check_case(seed):
    from odtk import _C, box
    rotated, a, c, b, shapes, strides, thr, top_n, dtype, spread, bias_mode, table = case(seed)
    nb = 6 if rotated else 4
    g = torch.Generator().manual_seed(13000 + seed)
    cls_bias = (torch.randn(a * c, generator=g) * 1.5 - 3.0).cuda() if bias_mode in ('both', 'cls') else None
    box_bias = (torch.randn(a * nb, generator=g) * 0.3).cuda() if bias_mode == 'both' else None
    cls, box_h = [], []
    for h, w in shapes:
        cls.append((torch.randn(b, a * c, h, w, generator=g) * spread - (0.0 if cls_bias is not None else 3.0)).to(dtype).cuda().contiguous(memory_format=torch.channels_last))
        box_h.append((torch.randn(b, a * nb, h, w, generator=g) * 0.3).to(dtype).cuda().contiguous(memory_format=torch.channels_last))
    if rotated:
        anchors = {s: box.generate_anchors_rotated(s, [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)], [-math.pi / 6, 0, math.pi / 6])[0] for s in set(strides)}
    else:
        anchors = {s: box.generate_anchors(s, [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)])[:a].contiguous() for s in set(strides)}
    alist = [anchors[s] for s in strides]
    kw = {}
    if cls_bias is not None:
        kw['cls_bias'] = cls_bias
        if table:
            kw['cls_thresholds'] = _C.prefilter_thresholds(cls_bias, dtype, thr)
    if box_bias is not None:
        kw['box_bias'] = box_bias
    got = _C.decode_levels(cls, box_h, alist, strides, thr, top_n, rotated, return_indices=True, logits=True, **kw)
    zc = torch.zeros(1, device='cuda')
    scores = [(x.float() + (cls_bias.view(1, -1, 1, 1) if cls_bias is not None else zc.view(1, 1, 1, 1))).sigmoid().to(dtype).float().contiguous() for x in cls]
    deltas = [(x.float() + (box_bias.view(1, -1, 1, 1) if box_bias is not None else zc.view(1, 1, 1, 1))).contiguous() for x in box_h]
    ref = _C.decode_levels(scores, deltas, alist, strides, thr, top_n, rotated, return_indices=True)
    for name, x, y in zip(('scores', 'boxes', 'classes', 'indices'), got, ref):
        if not torch.equal(x, y):
            return '%s differ (%d of %d entries)' % (name, int((x != y).sum()), x.numel())
    return ''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', default='0:3000', help='lo:hi, or a comma-separated list')
    args = ap.parse_args()
    seeds = [int(v) for v in args.seeds.split(',')] if ',' in args.seeds else list(range(*(int(v) for v in args.seeds.split(':'))))
    t0 = time.time()
    bad, n_cand = [], 0
    for seed in seeds:
        try:
            why = check_case(seed)
        except Exception as e:
            why = 'exception: %s' % str(e)[:300]
        if why:
            bad.append((seed, why))
            print('seed %d: MISMATCH %s   case %s' % (seed, why, case(seed)), flush=True)
    print('%d cases (seeds %d..%d) in %.0f s: %d mismatches%s' % (len(seeds), seeds[0], seeds[-1], time.time() - t0, len(bad), (' ' + str(bad[:10])) if bad else ''))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
