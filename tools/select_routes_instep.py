#!/usr/bin/env python
"""Which route select_decode's shared segments take WHERE THE KERNEL ACTUALLY RUNS: inside Model.forward of the
bench configuration -- level streams on, the head towers of P4..P7 on side streams next to P3's -- traced step by step
(odtk_debug_set_trace): per segment with G > 1 partners the route (cooperative / tournament), and for workgroup 0 of the segment
how long it sat between "slice fetched" and "barrier passed" (the segment's histogram atomics + the bounded spin on the ticket
counter).  Compared with the same launch stand-alone (box.detect on the captured head tensors, nothing else on the chip), and
with ODTK_SELECT_COOP_TICKS lowered (the environment of THIS process decides: run the script once per setting).

    python tools/select_routes_instep.py [--steps 200] > profiles/r06_select_routes_instep.txt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch  # noqa: E402

torch.backends.cudnn.benchmark = True
from odtk import _C, box  # noqa: E402
from odtk.model import Model  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--batch', type=int, default=8)
args = ap.parse_args()

torch.manual_seed(0)
m = Model('ResNet50FPN')
m.initialize(None)
m = m.cuda().to(memory_format=torch.channels_last).eval()
B = args.batch
x = torch.randn(B, 3, 800, 1280, generator=torch.Generator().manual_seed(0)).cuda().contiguous(memory_format=torch.channels_last)
eng = lambda: m.inference_engine(torch.bfloat16)
bench.calibrate_cls_head(m, lambda t: eng().heads(t), x, bench.SPEC_FRACTION, m.threshold)
with torch.no_grad():
    cls, dl = eng().heads(x)
strides = [8, 16, 32, 64, 128]
for s in strides:
    m.level_anchors(s)


def step():
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        return m(x)


def alone():
    return box.detect(cls, dl, strides, m.anchors, 0.05, 1000, 0.5, 100, False, logits=True)


def survey(fn, steps):
    trace = torch.zeros(_C.TRACE_WORDS, dtype=torch.int64, device='cuda')
    for _ in range(10):
        fn()
    shared = coop = 0
    waits, fetch, total = [], [], []
    per_level = {}
    for _ in range(steps):
        trace.zero_()
        _C.debug_set_trace(trace)
        fn()
        torch.cuda.synchronize()
        _C.debug_set_trace(None)
        t = trace.cpu()
        seg = t[:8192].view(-1, 8)[:5 * B]
        fine = t[8192:8192 + 64 * 16].view(-1, 16)[:5 * B]
        for i in range(5 * B):
            word = int(seg[i][7])
            g, was_coop = (word >> 1) & 0x7fff, (word >> 16) & 1
            if g <= 1:
                continue
            shared += 1
            coop += was_coop
            lvl = per_level.setdefault(i // B, [0, 0])
            lvl[0] += 1
            lvl[1] += was_coop
            f = fine[i]
            if int(f[2]) and int(f[4]):
                waits.append((int(f[4]) - int(f[2])) / 100.0)
                fetch.append((int(f[2]) - int(f[0])) / 100.0)
            total.append((int(seg[i][4]) - int(seg[i][0])) / 100.0)
    q = lambda v, p: sorted(v)[min(len(v) - 1, int(p * len(v)))] if v else float('nan')
    return {'steps': steps, 'shared_segments': shared, 'cooperative': coop, 'timed_out_or_vetoed': shared - coop,
            'per_level (segments, cooperative)': {('P%d' % (k + 3)): v for k, v in per_level.items()},
            'wg0: slice fetched -> barrier passed, us (p50, p90, p99, max)': [round(q(waits, p), 2) for p in (0.5, 0.9, 0.99)] + [round(max(waits), 2) if waits else None],
            'wg0: entry -> slice fetched, us (p50, p99)': [round(q(fetch, 0.5), 2), round(q(fetch, 0.99), 2)],
            'last finisher: lengths read -> done, us (p50, p99)': [round(q(total, 0.5), 2), round(q(total, 0.99), 2)]}


print('ODTK_SELECT_COOP_TICKS =', os.environ.get('ODTK_SELECT_COOP_TICKS', '(default 3000 = 30 us)'), ' ODTK_SELECT_RANK =',
      os.environ.get('ODTK_SELECT_RANK', '(default 1)'))
for name, fn in (('in Model.forward (level streams on)', step), ('stand-alone box.detect', alone)):
    r = survey(fn, args.steps)
    print('== %s' % name)
    for k, v in r.items():
        print('   %-68s %s' % (k, v))
_C.profile_enable(True, ('select_decode_kernel', 'nms_kernel', 'prefilter_scan_kernel'))
_C.profile_collect()
for _ in range(50):
    step()
torch.cuda.synchronize()
print('in-step kernel times (event-timed, 50 steps):', {k: round(v[0] / v[1] * 1e3, 2) for k, v in _C.profile_collect().items() if v[1]})
_C.profile_enable(False)
