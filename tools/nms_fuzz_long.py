#!/usr/bin/env python
"""Long differential fuzz of the axis-aligned NMS on DETECTOR-LIKE candidates -- clusters of overlapping same-class boxes around a few
objects, scores tying on a 16-bit grid, a handful of dominant classes -- in both input forms: arbitrary order (odtk_nms_ex) and
sorted runs (odtk_nms_sorted_runs, what odtk_detect hands over).  Every case is compared bit for bit with the C restatement of the
reference's CPU nms (oracle/c/odtk_oracle.c, canonical tie rule), and the two forms with each other.  Round 6's rounds -- batched
pushes, the capped push over everything and its bail-out, filter passes -- are taken or not depending on the case's yield, which
is what the generator varies.  tests/test_gpu_fuzz.py runs the first seeds of the same generator.

    python tools/nms_fuzz_long.py --seeds 0:400
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


def clustered_case(seed):
    """-> (scores [b, count], boxes [b, count, 4], classes [b, count], run_len, thr, ndet): n_runs runs, each in NMS order."""
    r = np.random.default_rng(9000 + seed)
    b = int(r.integers(1, 5))
    n_runs = int(r.choice([1, 2, 5, 5, 5, 6]))
    run_len = int(r.choice([64, 100, 300, 1000, 1000, 1500]))
    if n_runs * run_len > 9000:
        run_len = 1000
    count = n_runs * run_len
    ndet = int(r.choice([1, 5, 100, 100, 100, 300, 300]))
    thr = float(r.choice([0.0, 0.3, 0.5, 0.5, 0.7]))
    n_cls = int(r.choice([1, 2, 6, 6, 80]))
    n_obj = int(r.choice([1, 3, 20, 60, 400]))
    size = float(r.choice([128.0, 512.0, 1280.0]))
    jitter = float(r.choice([0.02, 0.08, 0.3]))             # of the object's size: how tight the clusters are
    fill = float(r.choice([0.1, 0.5, 0.9, 1.0]))            # share of a run's slots that hold a candidate
    quant = str(r.choice(['none', 'bf16', 'coarse']))
    g = torch.Generator().manual_seed(7000 + seed)
    obj_ctr = torch.rand(b, n_obj, 2, generator=g) * size
    obj_wh = torch.rand(b, n_obj, 2, generator=g) * size * 0.25 + 8
    obj_cls = torch.randint(0, n_cls, (b, n_obj), generator=g)
    which = torch.randint(0, n_obj, (b, count), generator=g)
    ctr = torch.gather(obj_ctr, 1, which[..., None].expand(-1, -1, 2))
    wh = torch.gather(obj_wh, 1, which[..., None].expand(-1, -1, 2))
    ctr = ctr + (torch.rand(b, count, 2, generator=g) - 0.5) * wh * jitter * 2
    wh = wh * (1 + (torch.rand(b, count, 2, generator=g) - 0.5) * jitter * 2)
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 2).contiguous()
    classes = torch.gather(obj_cls, 1, which)
    stray = torch.rand(b, count, generator=g) < 0.1          # a detector's confusions: another class on the same object
    classes = torch.where(stray, torch.randint(0, n_cls, (b, count), generator=g), classes).float()
    scores = torch.rand(b, count, generator=g) * 0.95 + 0.05
    if quant == 'bf16':
        scores = scores.bfloat16().float()
    elif quant == 'coarse':
        scores = (scores * 16).round() / 16
    scores[torch.rand(b, count, generator=g) >= fill] = 0
    # runs in NMS order: score descending inside every run, the empty slots (score 0) behind; position ascending breaks ties by construction
    order = torch.argsort(scores.view(b, n_runs, run_len), dim=2, descending=True, stable=True) + (torch.arange(n_runs) * run_len)[None, :, None]
    order = order.view(b, count)
    scores = torch.gather(scores, 1, order).contiguous()
    boxes = torch.gather(boxes, 1, order[..., None].expand(-1, -1, 4)).contiguous()
    classes = torch.gather(classes, 1, order).contiguous()
    return scores, boxes, classes, run_len, thr, ndet


def check_case(seed):
    """'' or what differs.  (import inside: the module is also imported by the CPU-side collection of tests/test_gpu_fuzz.py)"""
    from oracle import c_oracle
    from odtk import _C
    scores, boxes, classes, run_len, thr, ndet = clustered_case(seed)
    dev = [t.cuda() for t in (scores, boxes, classes)]
    gen = _C.nms(*dev, thr, ndet, False, return_indices=True)
    ref = c_oracle.nms(scores.numpy(), boxes.numpy(), classes.numpy(), thr, ndet, rotated=False)
    if not np.array_equal(gen[3].cpu().numpy().astype(np.int64), ref[3]):
        return 'generic form: kept positions differ from the oracle'
    for name, h, e in zip(('scores', 'boxes', 'classes'), gen[:3], ref[:3]):
        if not np.array_equal(np.ascontiguousarray(h.cpu().numpy()).view(np.uint32), e.view(np.uint32)):
            return 'generic form: %s differ from the oracle' % name
    runs = _C.nms_sorted_runs(*dev, run_len, thr, ndet)
    for name, x, y in zip(('scores', 'boxes', 'classes'), runs, gen[:3]):
        if not torch.equal(x, y):
            return 'sorted-run form: %s differ from the generic form' % name
    return ''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', default='0:400', help='lo:hi')
    args = ap.parse_args()
    lo, hi = (int(v) for v in args.seeds.split(':'))
    t0 = time.time()
    bad = []
    kept_hist = {}
    for seed in range(lo, hi):
        why = check_case(seed)
        if why:
            bad.append((seed, why))
            print('seed %d: %s   case %s' % (seed, why, [tuple(t.shape) if hasattr(t, 'shape') else t for t in clustered_case(seed)]), flush=True)
    print('%d cases (seeds %d..%d) in %.0f s: %d mismatches%s' % (hi - lo, lo, hi - 1, time.time() - t0, len(bad), (' ' + str(bad[:10])) if bad else ''))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
