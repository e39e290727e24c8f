#!/usr/bin/env python
"""Where the time of the file-based loader goes (odtk/data.py): first-batch latency (worker start-up) vs steady rate,
for several worker counts, on N synthetic 1280x800 JPEGs; plus the per-stage cost in the main process."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import numpy as np
import torch
from PIL import Image

from odtk.data import CocoDataset, DataIterator, normalise_batch

N = int(os.environ.get('PROBE_IMAGES', 192))
scratch = tempfile.mkdtemp(prefix='odtk_probe_')
yy, xx = np.mgrid[0:800, 0:1280]
images = []
for k in range(N):
    base = np.stack([(xx + 3 * k) % 256, (yy * 2 + k) % 256, ((xx + yy) // 2) % 256], 2).astype(np.uint8)
    Image.fromarray(base, 'RGB').save(os.path.join(scratch, 'im%04d.jpg' % k), quality=90)
    images.append({'id': k, 'file_name': 'im%04d.jpg' % k, 'width': 1280, 'height': 800})
ann = os.path.join(scratch, 'ann.json')
json.dump({'images': images}, open(ann, 'w'))
cuda = torch.cuda.is_available()
sync = torch.cuda.synchronize if cuda else (lambda: None)
if cuda:
    torch.zeros(1, device='cuda')
print('host cores', os.cpu_count(), 'images', N)

ds = CocoDataset(scratch, 800, 1333, 128, ann)
t = time.time(); items = [ds[i] for i in range(16)]; per_item = (time.time() - t) / 16
t = time.time(); packed = [ds.collate_fn(items[:8]) for _ in range(4)][0][0]; per_collate = (time.time() - t) / 4
dev = packed.cuda() if cuda else packed
sync(); t = time.time()
for _ in range(10):
    out = normalise_batch(packed.cuda() if cuda else packed)
sync(); per_norm = (time.time() - t) / 10
print('main process: item (decode + resize + to uint8) %.1f ms, collate of 8 %.1f ms, upload + normalise of 8 %.2f ms'
      % (per_item * 1e3, per_collate * 1e3, per_norm * 1e3))

for workers in [int(w) for w in os.environ.get('PROBE_WORKERS', '0,4,16,32').split(',')]:
    it = DataIterator(scratch, 800, 1333, 8, 128, 1, ann, training=False, num_workers=workers)
    t0 = time.time()
    first, n = None, 0
    for data, ids, ratios in it:
        n += data.shape[0]
        if first is None:
            sync()
            first = time.time() - t0
    sync()
    total = time.time() - t0
    print('%2d workers: first batch after %.2f s, then %.1f img/s; whole pass %.1f img/s'
          % (workers, first, (n - 8) / max(total - first, 1e-9), n / total))
