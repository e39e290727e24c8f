#!/usr/bin/env python
"""Corners of the NMS kernel that the ABI allows and the test-suite does not visit yet (round-3 first run):
detections_per_im above the 1024 candidates of one round (up to ODTK_MAX_NMS_DETECTIONS = 2048), with few and with very
many survivors, LDS-resident and workspace key lists, axis-aligned and rotated.  Compares with the C oracle; prints one line
per case and exits non-zero on the first mismatch.  Promote the cases to tests/test_gpu_fuzz.py once they have run green."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import numpy as np
import torch

from oracle import c_oracle
from odtk import _C

bad = 0
for seed, (count, ndet, thr, n_cls, rotated, spread) in enumerate([
        (5000, 1000, 0.5, 80, False, 600.0), (5000, 2048, 0.5, 80, False, 2000.0), (7680, 2048, 0.9, 1, False, 3000.0),
        (2048, 2048, 1.0, 2, False, 100.0), (10000, 1500, 0.5, 80, False, 1500.0), (3000, 1025, 0.3, 4, False, 400.0),
        (2500, 1200, 0.5, 3, True, 800.0), (1500, 2048, 0.7, 1, True, 300.0)]):
    g = torch.Generator().manual_seed(900 + seed)
    b = 2
    ctr = torch.rand(b, count, 2, generator=g) * spread + 20
    wh = torch.rand(b, count, 2, generator=g) * 40 + 2
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 2)
    if rotated:
        th = (torch.rand(b, count, generator=g) - 0.5) * 3.0
        boxes = torch.cat([boxes, th.sin()[..., None], th.cos()[..., None]], 2)
    scores = torch.rand(b, count, generator=g)
    if seed % 2:
        scores = (scores * 64).round() / 64
    classes = torch.randint(0, n_cls, (b, count), generator=g).float()
    out = _C.nms(scores.cuda(), boxes.cuda(), classes.cuda(), thr, ndet, rotated, return_indices=True)
    ref = c_oracle.nms(scores.numpy(), boxes.numpy(), classes.numpy(), thr, ndet, rotated=rotated)
    same = np.array_equal(out[3].cpu().numpy().astype(np.int64), ref[3]) and all(
        np.array_equal(np.ascontiguousarray(h.cpu().numpy()).view(np.uint32), e.view(np.uint32)) for h, e in zip(out[:3], ref[:3]))
    kept = int((out[0] > 0).sum(1).max())
    print('count %5d ndet %4d thr %.1f classes %2d rotated %-5s: kept up to %4d  %s' % (count, ndet, thr, n_cls, rotated, kept,
                                                                                       'OK' if same else 'MISMATCH'), flush=True)
    bad += not same
sys.exit(1 if bad else 0)
