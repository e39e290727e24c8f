#!/usr/bin/env python
"""Effective shader clock seen by the post-processing kernels: in the inference step (right behind ~7 ms of
convolutions) vs back to back on an otherwise idle chip.  The NMS kernel's debug trace records shader cycles
(s_memtime) next to the constant 100 MHz wall clock; cycles / wall time = the clock the kernel actually ran at.
Explains why latency-bound launches measure ~1.35x longer inside bench.py's step than in tools/postproc_bench.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
torch.backends.cudnn.benchmark = True
from odtk import _C, box
from odtk.model import Model
import bench

torch.manual_seed(0)
m = Model('ResNet50FPN'); m.initialize(None)
m = m.cuda().to(memory_format=torch.channels_last).eval()
x = torch.randn(8, 3, 800, 1280, generator=torch.Generator().manual_seed(0)).cuda().contiguous(memory_format=torch.channels_last)
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


def clock_of(fn, warm=10, reps=5):
    for _ in range(warm):
        fn()
    out = []
    for _ in range(reps):
        trace = torch.zeros(_C.TRACE_WORDS, dtype=torch.int64, device='cuda')
        _C.debug_set_trace(trace)
        fn(); torch.cuda.synchronize()
        _C.debug_set_trace(None)
        t = trace.cpu().view(-1, 8)[64 + 8:64 + 16]          # the 8 nms workgroups
        wall_us = (t[:, 4] - t[:, 0]).float() / 100.0
        ghz = t[:, 7].float() / (wall_us * 1e3)
        out.append((float(wall_us.mean()), float(ghz.mean())))
        for _ in range(3):
            fn()
    return out


print('in the inference step : nms wall us, effective GHz', clock_of(step))
print('back to back, idle chip: nms wall us, effective GHz', clock_of(alone))


# per-chunk timeline of image 0 (in the step)
for _ in range(3):
    step()
trace = torch.zeros(_C.TRACE_WORDS, dtype=torch.int64, device='cuda')
_C.debug_set_trace(trace)
step(); torch.cuda.synchronize()
_C.debug_set_trace(None)
t = trace.cpu()
rows = t[4096:4096 + 80].view(-1, 4)
img0 = t.view(-1, 8)[64 + 8]
print('nms image 0: compact %.2f | select %.2f | sort %.2f | chunks %.2f us; consumed %d of %d' % (
    (img0[1] - img0[0]) / 100.0, (img0[2] - img0[1]) / 100.0, (img0[3] - img0[2]) / 100.0, (img0[4] - img0[3]) / 100.0, img0[5], img0[6]))
prev = None
for c, r in enumerate(rows):
    if int(r[0]) == 0:
        break
    print('  chunk %2d: pull %.2f us | resolve %.2f us | kept after %d%s' % (c, (r[1] - r[0]) / 100.0, (r[2] - r[1]) / 100.0, int(r[3]),
          '' if prev is None else ' | gap before %.2f' % ((r[0] - prev) / 100.0)))
    prev = int(r[2])
