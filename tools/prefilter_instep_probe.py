#!/usr/bin/env python
"""Why is the prefilter slower inside the step than back to back?  Same engine, same head tensors, four contexts:
  step        heads -> detect                                  (what bench.py times)
  twice       heads -> detect -> detect                        (the second launch: same data, same clocks, nothing in between)
  idle        heads -> ~60 us of idle GPU (torch.cuda._sleep) -> detect
  alone       detect only, back to back on the head tensors of the last step
Prints the average prefilter / select_decode / nms launch time per context (library event hooks)."""
import os
import sys
import json
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
bench.calibrate_cls_head(m, lambda t: m.inference_engine(torch.bfloat16).heads(t), x, bench.SPEC_FRACTION, m.threshold)
e = m.inference_engine(torch.bfloat16)
strides = [8, 16, 32, 64, 128]
for s in strides:
    m.level_anchors(s)


def detect(cls, dl, cb, bb):
    return box.detect(cls, dl, strides, m.anchors, m.threshold, m.top_n, m.nms, m.detections, False, logits=True, cls_bias=cb, box_bias=bb)


def run(kind, n=20):
    post = ('prefilter_scan_kernel', 'select_decode_kernel', 'nms_kernel')
    with torch.no_grad():
        cls, dl, cb, bb = e.heads_without_last_bias(x)
        for _ in range(3):
            detect(cls, dl, cb, bb)
        torch.cuda.synchronize()
        _C.profile_enable(True, post)
        _C.profile_collect()
        for _ in range(n):
            if kind != 'alone':
                cls, dl, cb, bb = e.heads_without_last_bias(x)
            if kind == 'idle':
                torch.cuda._sleep(150000)
            detect(cls, dl, cb, bb)
            if kind == 'twice':
                detect(cls, dl, cb, bb)
        torch.cuda.synchronize()
        _C.profile_enable(False)
        prof = _C.profile_collect()
    return {k: round(prof[k][0] / max(prof[k][1], 1) * 1e3, 2) for k in post}


res = {}
for kind in ('step', 'twice', 'idle', 'alone', 'step'):
    r = run(kind)
    res.setdefault(kind, []).append(r)
    print(kind, r, flush=True)
t1 = res['step'][0]['prefilter_scan_kernel']
t12 = res['twice'][0]['prefilter_scan_kernel']
print('second launch of "twice" (2 x avg - first):', round(2 * t12 - t1, 2))
print(json.dumps(res))
