#!/usr/bin/env python
"""Where does the bf16 engine lose its ~1.2 AP points?  The setup of tests/test_gpu_detection_parity.py
(RN50FPN, 800x1280, ~45 planted objects per image, truth = the fp32 eager graph + oracle post-processing), the acceptance
metric (COCO AP, odtk/cocoeval.py), and the engine taken apart stage by stage:

  engine_fp32                     the BN-folded engine in fp32                                   (the detector itself)
  weights_bf16_act_fp32           fp32 arithmetic / activations, every folded weight rounded to bf16
  backbone_fp32_heads_bf16        fp32 backbone + FPN, the head towers and the post-processing in bf16
  backbone_bf16_heads_fp32        bf16 backbone + FPN, the head towers in fp32
  engine_bf16_logits_fp32         bf16 everywhere, the LAST cls / box convolutions accumulate and store fp32
  engine_bf16_logits_fp16         bf16 everywhere, the last convolutions' fp32 accumulators stored as fp16
  engine_bf16                     the timed path (logits stored as bf16)
  engine_fp16                     what `odtk infer` runs by default

Prints one AP per line; run on the GPU box:  python tools/bf16_ablation.py > profiles/r04_bf16_ablation.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import test_gpu_detection_parity as T  # noqa: E402
from odtk import box  # noqa: E402
from odtk.fused import FusedRetinaNet  # noqa: E402

torch.backends.cudnn.benchmark = True
model, x = T.build_model()
ref = T.reference_detections(model, x)
planted = ref[0] >= 0.15
truth = (ref[0] * planted, ref[1] * planted[..., None], ref[2] * planted)
strides = [8, 16, 32, 64, 128]
for s in strides:
    model.level_anchors(s)


def detect(cls, bx, cls_bias=None, box_bias=None):
    return box.detect(cls, bx, strides, model.anchors, model.threshold, model.top_n, model.nms, model.detections, False,
                      logits=True, cls_bias=cls_bias, box_bias=box_bias)


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def towers_with_wide_last_conv(eng, feats, store):
    """Head towers of `eng` (bf16) with the last convolutions evaluated in fp32 on the bf16 activations and stored as `store`."""
    cls_out, box_out = [], []
    for t in feats:
        c = eng._run(eng.cls_head[:-1], t)
        b = eng._run(eng.box_head[:-1], t)
        lc, lb = eng.cls_head[-1], eng.box_head[-1]
        c32 = F.conv2d(c.float(), lc.weight.float(), None, lc.stride, lc.padding) + lc.bias.view(1, -1, 1, 1)
        b32 = F.conv2d(b.float(), lb.weight.float(), None, lb.stride, lb.padding) + lb.bias.view(1, -1, 1, 1)
        cls_out.append(cl(c32.to(store)))
        box_out.append(cl(b32.to(store)))
    return cls_out, box_out


results = {}
with torch.no_grad():
    e32 = FusedRetinaNet(model, torch.float32).cuda()
    e16 = FusedRetinaNet(model, torch.bfloat16).cuda()
    results['engine_fp32'] = e32(x)
    results['engine_bf16'] = e16(x)
    results['engine_fp16'] = FusedRetinaNet(model, torch.float16).cuda()(x)
    # fp32 arithmetic, bf16-rounded weights
    ew = FusedRetinaNet(model, torch.float32).cuda()
    for name, buf in ew.named_buffers():
        if name.endswith('weight'):
            buf.copy_(buf.bfloat16().float())
    results['weights_bf16_act_fp32'] = ew(x)
    # fp32 backbone + FPN -> bf16 towers
    xc = cl(x.float())
    feats32 = e32.features(xc)
    c, b = e16._towers([cl(f.bfloat16()) for f in feats32], True)
    results['backbone_fp32_heads_bf16'] = detect(c, b)
    # bf16 backbone + FPN -> fp32 towers
    feats16 = e16.features(cl(x.bfloat16()))
    c, b = e32._towers([cl(f.float()) for f in feats16], True)
    results['backbone_bf16_heads_fp32'] = detect(c, b)
    # bf16 everywhere, last convolutions wide
    for store, key in ((torch.float32, 'engine_bf16_logits_fp32'), (torch.float16, 'engine_bf16_logits_fp16'),
                       (torch.bfloat16, 'engine_bf16_logits_bf16_recomputed')):
        c, b = towers_with_wide_last_conv(e16, feats16, store)
        results[key] = detect(c, b)

print('COCO AP (IoU 0.50:0.95) against the planted objects of the fp32 reference pipeline; RN50FPN 800x1280, batch %d' % x.shape[0])
print('%-36s %8.4f' % ('reference (fp32 eager + oracle)', T.coco_ap(truth, ref)))
for k in ('engine_fp32', 'weights_bf16_act_fp32', 'backbone_fp32_heads_bf16', 'backbone_bf16_heads_fp32', 'engine_bf16_logits_fp32',
          'engine_bf16_logits_fp16', 'engine_bf16_logits_bf16_recomputed', 'engine_bf16', 'engine_fp16'):
    a = T.agreement(ref, results[k], 0.08, min_iou=0.5)
    print('%-36s %8.4f   max |dscore| %.4f  matched %d / %d' % (k, T.coco_ap(truth, results[k]), a['max_dscore'], a['matched'], a['eligible']))
