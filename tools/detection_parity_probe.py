#!/usr/bin/env python
"""Prints the agreement statistics tests/test_gpu_detection_parity.py pins, for a sweep of score margins."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import torch  # noqa: E402
import test_gpu_detection_parity as T  # noqa: E402

for ridge in (1e-2,):
    model, x = T.build_model(ridge=ridge)
    ref = T.reference_detections(model, x)
    paths = T.candidate_paths(model, x)
    print('=== ridge', ridge, 'reference detections', int((ref[0] > 0).sum()), '>=0.15:', int((ref[0] >= 0.15).sum()),
          '|w| of the fitted layer %.3f' % float(model.cls_head[-1].weight.abs().max()))
    print('reference scores image 0:', [round(float(v), 3) for v in ref[0][0][:60]])
    with torch.no_grad():
        ref_cls, _ = model.heads(x)
        eng = model.inference_engine(torch.bfloat16)
        eng_cls, _ = eng.heads(x)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            ea_cls, _ = model.heads(x)
    for name, cl in (('engine_bf16', eng_cls), ('eager_autocast', ea_cls)):
        r = torch.cat([c.flatten() for c in ref_cls])
        g = torch.cat([c.float().flatten() for c in cl])
        hot = r > -3.0
        print(name, 'max |dlogit| %.4f' % (g - r).abs().max().item(), 'rms %.5f' % (g - r).pow(2).mean().sqrt().item(),
              'on logits > -3: amplitude ratio %.4f' % ((g[hot] + 4.595).mean() / (r[hot] + 4.595).mean()).item())
    for name, got in paths.items():
        for margin in (2e-4, 1e-3, 3e-3, 1e-2, 2e-2, 4e-2, 8e-2, 0.15, 0.3):
            print(name, margin, 'IoU>=0.9', json.dumps(T.agreement(ref, got, margin)), 'IoU>=0.5', json.dumps(T.agreement(ref, got, margin, min_iou=0.5)))
