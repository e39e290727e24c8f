#!/usr/bin/env python
"""Per-image trace of the NMS launch on the head tensors of a bench configuration (odtk_debug_set_trace): candidates with a
positive score (K), candidates examined (consumed), time per phase, effective shader clock -- inside Model.forward and back to
back -- plus the per-chunk timeline of image 0.  Also saves decode_levels' output of the batch (the NMS input) to
gpurun_out/nms_inputs_<tag>.pt so that the pair statistics can be studied on the host.

    python tools/nms_trace_probe.py --backbone ResNet101FPN --batch 16
    python tools/nms_trace_probe.py --rotated-bbox [--unit-rotation]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch

torch.backends.cudnn.benchmark = True
from odtk import _C, box
from odtk.model import Model
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--backbone', default='ResNet50FPN')
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--rotated-bbox', action='store_true')
ap.add_argument('--unit-rotation', action='store_true')
ap.add_argument('--tag', default=None)
args = ap.parse_args()
tag = args.tag or '%s_bs%d%s%s' % (args.backbone, args.batch, '_rot' if args.rotated_bbox else '', '_unit' if args.unit_rotation else '')

torch.manual_seed(0)
m = Model(args.backbone, rotated_bbox=args.rotated_bbox)
m.initialize(None)
if args.rotated_bbox and args.unit_rotation:
    with torch.no_grad():
        b = m.box_head[-1].bias.view(m.num_anchors, 6)
        b.zero_()
        b[:, 5] = 1.0
m = m.cuda().to(memory_format=torch.channels_last).eval()
x = torch.randn(args.batch, 3, 800, 1280, generator=torch.Generator().manual_seed(0)).cuda().contiguous(memory_format=torch.channels_last)
eng = lambda: m.inference_engine(torch.bfloat16)
bench.calibrate_cls_head(m, lambda t: eng().heads(t), x, bench.SPEC_FRACTION, m.threshold)
with torch.no_grad():
    cls, dl = eng().heads(x)
strides = [8, 16, 32, 64, 128]
for s in strides:
    m.level_anchors(s)
B = args.batch


def step():
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        return m(x)


def alone():
    return box.detect(cls, dl, strides, m.anchors, 0.05, 1000, 0.5, 100, args.rotated_bbox, logits=True)


def traced(fn):
    trace = torch.zeros(_C.TRACE_WORDS, dtype=torch.int64, device='cuda')
    _C.debug_set_trace(trace)
    out = fn()
    torch.cuda.synchronize()
    _C.debug_set_trace(None)
    return trace.cpu(), out


for name, fn in (('in Model.forward', step), ('back to back', alone)):
    for _ in range(8):
        fn()
    t, out = traced(fn)
    rows = t.view(-1, 8)[64 + B:64 + 2 * B]
    t0 = int(rows[:, 0].min())
    print('== %s (%s): nms workgroups, us relative to the first start' % (tag, name))
    for i, r in enumerate(rows):
        wall = (int(r[4]) - int(r[0])) / 100.0
        print('  img %2d: start %6.2f | setup %5.2f | round1 load/select %5.2f | order %5.2f | chunks %7.2f | total %7.2f us | consumed %5d of K %5d | kept %3d | %.2f GHz'
              % (i, (int(r[0]) - t0) / 100.0, (int(r[1]) - int(r[0])) / 100.0, (int(r[2]) - int(r[1])) / 100.0,
                 (int(r[3]) - int(r[2])) / 100.0, (int(r[4]) - int(r[3])) / 100.0, wall, int(r[5]), int(r[6]),
                 int((out[0][i] > 0).sum()), int(r[7]) / max(wall * 1e3, 1e-9)))
    ph = t[4096 + 96:4096 + 96 + 96].view(-1, 2)
    names = {1: 'round selected', 2: 'boxes staged', 3: 'chunks / push done', 4: 'filter done', 5: 'push over everything done',
             10: 'batch compacted', 11: 'rows', 12: 'resolved', 13: 'pushed', 20: 'partitioned by class', 21: 'classes resolved'}
    line, prev_t = [], int(rows[0][0])
    for pid, pt in ph.tolist():
        if pid == 0:
            break
        line.append('%s +%.2f' % (names.get(pid, str(pid)), (pt - prev_t) / 100.0))
        prev_t = pt
    print('    img 0 phases (us since the previous one; first since kernel start): ' + ' | '.join(line))
    ch = t[4096:4096 + 80].view(-1, 4)
    prev = None
    for c, r in enumerate(ch):
        if int(r[0]) == 0:
            break
        print('    img 0 chunk %2d: pull+rows %.2f us | resolve %.2f us | kept after %d%s' % (
            c, (int(r[1]) - int(r[0])) / 100.0, (int(r[2]) - int(r[1])) / 100.0, int(r[3]),
            '' if prev is None else ' | gap before %.2f' % ((int(r[0]) - prev) / 100.0)))
        prev = int(r[2])

# kernel time of the launch itself, back to back
_C.profile_enable(True, ('nms_kernel', 'select_decode_kernel', 'prefilter_scan_kernel', 'select_hist_kernel', 'select_filter_kernel',
                         'nms_first_round_kernel', 'rotated_sup_matrix_kernel'))
_C.profile_collect()
for _ in range(20):
    alone()
torch.cuda.synchronize()
_C.profile_enable(False)
print('back to back, event-timed:', {k: round(v[0] / v[1] * 1e3, 2) for k, v in _C.profile_collect().items() if v[1]})

with torch.no_grad():
    bias = None
    dec = box.decode_levels(cls, dl, strides, 0.05, 1000, m.anchors, args.rotated_bbox, logits=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
torch.save([t.cpu() for t in dec], os.path.join(ROOT, 'gpurun_out', 'nms_inputs_%s.pt' % tag))
print('saved gpurun_out/nms_inputs_%s.pt' % tag)
