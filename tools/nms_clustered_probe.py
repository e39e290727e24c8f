#!/usr/bin/env python
"""The NMS launch on what a TRAINED detector hands it (tests/golden/nms_trained_scenes_ties.npz: 16 images, 100..2550 candidates in
clusters of overlapping same-class boxes, 16-bit scores tying in the hundreds) -- replayed through odtk_nms_sorted_runs, i.e. the
code path odtk_detect runs, without a model: per image candidates / examined / kept / time and the phase trace of image 0, the
launch's event time, and the result against the fixture's expected outputs (canonical rule) bit for bit.

    python tools/nms_clustered_probe.py [--fixture tests/golden/nms_trained_scenes_ties.npz] [--generic]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from odtk import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--fixture', default=os.path.join(ROOT, 'tests', 'golden', 'nms_trained_scenes_ties.npz'))
ap.add_argument('--generic', action='store_true', help='the stand-alone op (arbitrary order in) instead of the sorted-run form')
ap.add_argument('--images', type=int, default=0, help='first N images only (0 = all)')
ap.add_argument('--first', type=int, default=0, help='put this image first (the phase trace is image 0\'s)')
args = ap.parse_args()
g = np.load(args.fixture)
n = args.images or g['scores'].shape[0]
order = [args.first] + [i for i in range(g['scores'].shape[0]) if i != args.first]
order = order[:n]
scores, boxes, classes = (torch.from_numpy(g[k][order]).cuda() for k in ('scores', 'boxes', 'classes'))
nms, det = float(g['nms']), int(g['detections'])
B, count = scores.shape
run_len = count // 5


def run():
    if args.generic:
        return _C.nms(scores, boxes, classes, nms, det)
    return _C.nms_sorted_runs(scores, boxes, classes, run_len, nms, det)


for _ in range(5):
    out = run()
torch.cuda.synchronize()
ok = all(torch.equal(o.cpu(), torch.from_numpy(g[k][order])) for o, k in zip(out, ('out_scores', 'out_boxes', 'out_classes')))
print('result == fixture (canonical rule), bit for bit:', ok)
trace = torch.zeros(_C.TRACE_WORDS, dtype=torch.int64, device='cuda')
_C.debug_set_trace(trace)
run()
torch.cuda.synchronize()
_C.debug_set_trace(None)
t = trace.cpu()
rows = t.view(-1, 8)[64 + B:64 + 2 * B]
kept = (out[0] > 0).sum(1).tolist()
print('img | candidates | examined | kept | us: setup | round-1 loads | round-1 order | rest | total')
for i, r in enumerate(rows):
    us = lambda a, b: (int(r[b]) - int(r[a])) / 100.0
    print('%3d | %6d | %6d | %4d | %6.1f %6.1f %6.1f %7.1f | %7.1f' % (i, int(r[6]), int(r[5]), kept[i], us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(0, 4)))
ph = t[4096 + 96:4096 + 96 + 96].view(-1, 2)
names = {1: 'round selected', 2: 'boxes staged', 3: 'chunks / push done', 4: 'filter done', 5: 'push over everything done',
         10: 'batch compacted', 11: 'rows', 12: 'resolved', 13: 'pushed',
         20: 'partitioned by class', 21: 'classes resolved'}
line, prev_t = [], int(rows[0][0])
for pid, pt in ph.tolist():
    if pid == 0:
        break
    line.append('%s +%.1f' % (names.get(pid, str(pid)), (pt - prev_t) / 100.0))
    prev_t = pt
print('img 0 phases (us since the previous one): ' + ' | '.join(line))
ch = t[4096:4096 + 80].view(-1, 4)
for c, r in enumerate(ch):
    if int(r[0]) == 0:
        break
    print('   img 0 round 1 chunk %2d: pull+rows %.2f us | resolve %.2f us | kept after %d' % (c, (int(r[1]) - int(r[0])) / 100.0, (int(r[2]) - int(r[1])) / 100.0, int(r[3])))
_C.profile_enable(True, ('nms_kernel',))
_C.profile_collect()
for _ in range(50):
    run()
torch.cuda.synchronize()
_C.profile_enable(False)
print('launch, event-timed, back to back (50):', {k: round(v[0] / v[1] * 1e3, 2) for k, v in _C.profile_collect().items() if v[1]})
