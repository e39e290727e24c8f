#!/usr/bin/env python
"""Long differential fuzz of decode_levels (prefilter + select_decode) against the torch restatement of the reference's CPU decode
(oracle/box_oracle.py): the generator of tests/test_gpu_fuzz.py widened to levels of up to 128 x 160 cells x 9 anchors x 24 classes
(segments that several workgroups share: the cooperative route, its plateau fall-back, the tournament), thresholds down to 0,
top_n 1 .. 2000, fp32 scores as they are / rounded to bf16 (ties in the hundreds) / on a 1/8 grid (plateaus of thousands), and
the same inputs once more as bf16 channels_last tensors.  Indices, scores and classes bit for bit; boxes through
oracle/box_check.py (1e-4, or proven exp rounding).

    python tools/decode_fuzz_long.py --seeds 0:300
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]


def case(seed):
    r = np.random.default_rng(4000 + seed)
    a = int(r.integers(1, 10))
    c = int(r.choice([1, 3, 7, 12, 24]))
    levels = int(r.integers(1, 4))
    b = int(r.integers(1, 5))
    big = r.random() < 0.5
    shapes = [(int(r.integers(1, 128 if big else 40)), int(r.integers(1, 160 if big else 40))) for _ in range(levels)]
    # image extent <= 2560 px (tests/test_gpu_fuzz.py's bound; BASELINE's images are 800 x 1280): beyond ~2048 px ONE fp32 ulp of a
    # coordinate is 2.4e-4, more than twice the north star's 1e-4, and "which neighbour of the float64 truth" becomes a coin toss
    # between the reference's exp and the correctly rounded one (seed 9 of the first run: stride 64 x 73 cells, |got - truth| 1.11e-4
    # vs |ref - truth| 1.05e-4 on adjacent floats)
    strides = [int(r.choice([s for s in (4, 8, 16, 32, 64) if s * max(h, w) <= 2560] or [4])) for (h, w) in shapes]
    thr = float(r.choice([0.0, 0.02, 0.05, 0.3, 0.5, 0.9]))
    top_n = int(r.choice([1, 7, 64, 100, 1000, 1000, 1500, 2000]))
    spread = float(r.choice([0.5, 1.5, 4.0]))
    shift = float(r.choice([-3.0, -1.0, 0.0]))
    quant = str(r.choice(['none', 'bf16', 'coarse']))
    return a, c, b, shapes, strides, thr, top_n, spread, shift, quant


def check_case(seed):
    from oracle import box_check, box_oracle
    from odtk import _C, box
    a, c, b, shapes, strides, thr, top_n, spread, shift, quant = case(seed)
    g = torch.Generator().manual_seed(5000 + seed)
    cls, dl = [], []
    for (h, w) in shapes:
        s = (torch.randn(b, a * c, h, w, generator=g) * spread + shift).sigmoid()
        if quant == 'bf16':
            s = s.bfloat16().float()
        elif quant == 'coarse':
            s = (s * 8).round() / 8
        cls.append(s)
        dl.append(torch.randn(b, a * 4, h, w, generator=g) * 0.5)
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES)[:a].contiguous() for s in set(strides)}
    out = _C.decode_levels([x.cuda() for x in cls], [x.cuda() for x in dl], [anchors[s] for s in strides], strides, thr, top_n, False,
                           return_indices=True)
    ref = [box_oracle.decode(x, d, s, thr, top_n, anchors[s], return_indices=True) for x, d, s in zip(cls, dl, strides)]
    ref = [torch.cat(t, 1) for t in zip(*ref)]
    if not torch.equal(out[3].cpu().long(), ref[3]):
        return 'indices'
    if not (torch.equal(out[0].cpu(), ref[0]) and torch.equal(out[2].cpu(), ref[2])):
        return 'scores / classes'
    try:
        box_check.check_decode(out[1], ref[1], cls, dl, strides, anchors, thr, top_n, ref_indices=ref[3])
    except AssertionError as e:
        # the 1e-4 criterion (or its bounded exp-rounding escape) is not met: is the output at least the canonical arithmetic --
        # the C restatement (reference operation order, correctly rounded exp) bit for bit on EVERY coordinate -- and how large are
        # the boxes concerned (one fp32 ulp of a 1024..2048 px extent is 1.2e-4: above the tolerance)?
        from oracle import c_oracle
        exact = torch.cat([torch.from_numpy(c_oracle.decode(x.numpy(), d.numpy(), s, thr, top_n, anchors[s].numpy())[1]) for x, d, s in zip(cls, dl, strides)], 1)
        got = out[1].cpu()
        same = bool((got.view(torch.int32) == exact.view(torch.int32)).all())
        over = (got.double() - ref[1].double()).abs() > 1e-4
        ext = torch.maximum(ref[1][..., 2] - ref[1][..., 0], ref[1][..., 3] - ref[1][..., 1])[..., None].expand_as(over)
        return '%sboxes: %s | output == C restatement on every coordinate: %s; %d coordinates beyond 1e-4, on boxes of %.0f .. %.0f px (clamped extent), deltas |dw|,|dh| up to %.2f' % (
            'NOTE ' if same else '', str(e)[:160], same, int(over.sum()), float(ext[over].min()), float(ext[over].max()), max(float(d[:, 2::4].abs().max()) for d in dl))
    if quant == 'bf16':
        out16 = _C.decode_levels([x.cuda().bfloat16().contiguous(memory_format=torch.channels_last) for x in cls],
                                 [x.cuda().bfloat16().contiguous(memory_format=torch.channels_last) for x in dl],
                                 [anchors[s] for s in strides], strides, thr, top_n, False, return_indices=True)
        if not (torch.equal(out16[3], out[3]) and torch.equal(out16[0], out[0])):
            return 'bf16 channels_last tensors select differently'
    return ''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', default='0:300', help='lo:hi, or a comma-separated list')
    args = ap.parse_args()
    seeds = [int(v) for v in args.seeds.split(',')] if ',' in args.seeds else list(range(*(int(v) for v in args.seeds.split(':'))))
    lo, hi = seeds[0], seeds[-1] + 1
    t0 = time.time()
    bad, notes = [], []
    for seed in seeds:
        try:
            why = check_case(seed)
        except Exception as e:                               # (a refused input is a finding too: name the seed)
            why = 'exception: %s' % str(e)[:200]
        if why.startswith('NOTE '):
            # boxes beyond 1e-4 of the torch reference (or more of them than its exp-rounding escape allows per call) that ARE the
            # canonical arithmetic bit for bit: the reference's own exp (1 ulp off on 1.1 % of inputs) on boxes thousands of pixels wide
            notes.append(seed)
            print('seed %d: %s   case %s' % (seed, why[5:], case(seed)), flush=True)
        elif why:
            bad.append((seed, why))
            print('seed %d: MISMATCH %s   case %s' % (seed, why, case(seed)), flush=True)
    print('%d of the cases have box coordinates beyond 1e-4 of the torch reference while equal to the C restatement on every coordinate '
          '(wild deltas on the largest anchors: the reference\'s exp rounding): seeds %s' % (len(notes), notes))
    print('%d cases (seeds %d..%d) in %.0f s: %d mismatches%s' % (len(seeds), lo, hi - 1, time.time() - t0, len(bad), (' ' + str(bad[:10])) if bad else ''))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
