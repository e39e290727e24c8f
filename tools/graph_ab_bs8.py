#!/usr/bin/env python
"""Model.forward at the headline configuration (RN50FPN bf16 bs 8, 800x1280): eager launches vs the whole call replayed as ONE
hipGraph (Model.forward(x, graph=True)); three alternating rounds, wall time per step."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
torch.backends.cudnn.benchmark = True
from odtk.model import Model
import bench

torch.manual_seed(0)
m = Model('ResNet50FPN'); m.initialize(None)
m = m.cuda().to(memory_format=torch.channels_last).eval()
x = torch.randn(8, 3, 800, 1280, generator=torch.Generator().manual_seed(0)).cuda().contiguous(memory_format=torch.channels_last)
bench.calibrate_cls_head(m, lambda t: m.inference_engine(torch.bfloat16).heads(t), x, bench.SPEC_FRACTION, m.threshold)


def run(graph, n=30):
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        for _ in range(5):
            out = m(x, graph=graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = m(x, graph=graph)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, int((out[0] > 0).sum())


for r in range(3):
    e, d0 = run(False)
    g, d1 = run(True)
    print('round %d: eager %.3f ms/step (%.1f img/s, %d detections)   graph %.3f ms/step (%.1f img/s, %d detections)' % (r, e, 8e3 / e, d0, g, 8e3 / g, d1), flush=True)
