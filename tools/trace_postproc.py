#!/usr/bin/env python
"""Phase timing INSIDE select_decode / nms (debug trace, odtk_debug_set_trace) on the head tensors
of the calibrated bench model (real conv outputs: spatially correlated scores, bf16 ties)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
torch.backends.cudnn.benchmark = True
from odtk import _C, box
from odtk.model import Model
sys.path.insert(0, ROOT)
import bench

torch.manual_seed(0)
ROT = '--rotated' in sys.argv
m = Model('ResNet50FPN', rotated_bbox=ROT); m.initialize(None)
m = m.cuda().to(memory_format=torch.channels_last).eval()
x = torch.randn(8, 3, 800, 1280, generator=torch.Generator().manual_seed(0)).cuda().contiguous(memory_format=torch.channels_last)
bench.calibrate_cls_head(m, lambda t: m.inference_engine(torch.bfloat16).heads(t), x, bench.SPEC_FRACTION, m.threshold)
with torch.no_grad():
    cls, dl = m.inference_engine(torch.bfloat16).heads(x)
strides = [8, 16, 32, 64, 128]
for s in strides: m.level_anchors(s)
run = lambda: box.detect(cls, dl, strides, m.anchors, 0.05, 1000, 0.5, 100, ROT, logits=True)
for _ in range(3): run()
trace = torch.zeros(_C.TRACE_WORDS, dtype=torch.int64, device='cuda')
_C.debug_set_trace(trace)
run(); torch.cuda.synchronize()
_C.debug_set_trace(None)
t = trace.cpu()[:8192].view(-1, 8)
fine = trace.cpu()[8192:8192 + 64 * 16].view(-1, 16)
us = lambda a, b: (b - a).float() / 100.0
print('select_decode phases (us): read+select | lds narrow | sort | decode   [per level, mean over 8 images]')
for l in range(5):
    blk = t[l * 8:(l + 1) * 8]
    print('  P%d: %6.1f %6.1f %6.1f %6.1f   total %6.1f' % (l + 3, us(blk[:, 0], blk[:, 1]).mean(), us(blk[:, 1], blk[:, 2]).mean(),
          us(blk[:, 2], blk[:, 3]).mean(), us(blk[:, 3], blk[:, 4]).mean(), us(blk[:, 0], blk[:, 4]).mean()),
          'counts', blk[:, 5].tolist(), 'n_sort', blk[:, 6].tolist(), 'read+select per image', [round(float(v), 1) for v in us(blk[:, 0], blk[:, 1])])
print('select_decode, finer (us from kernel entry of workgroup 0; image 0 of each level): counts read | slice fetched | (refetched) | narrowed '
      '| published | ticket || finisher: ticket | count read | survivors fetched')
for l in range(5):
    r = fine[l * 8]
    rel = lambda k: ('%6.1f' % ((int(r[k]) - int(r[0])) / 100.0)) if int(r[k]) else '     -'
    print('  P%d: ' % (l + 3) + ' '.join(rel(k) for k in (1, 2, 3, 4, 5, 6)) + ' || ' + ' '.join(rel(k) for k in (8, 9, 10)),
          ' G =', (int(t[l * 8][7]) >> 1) & 0x7fff, '(cooperative route)' if (int(t[l * 8][7]) >> 16) & 1 else '')
n = t[64 + 8:64 + 16]
print('nms phases (us): compact | select round 0 | sort | chunks   consumed/K')
for b in range(8):
    r = n[b]
    print('  img%d: %6.1f %6.1f %6.1f %6.1f  total %6.1f   %d / %d' % (b, us(r[0], r[1]), us(r[1], r[2]), us(r[2], r[3]), us(r[3], r[4]), us(r[0], r[4]), r[5], r[6]))
