#!/usr/bin/env python
"""Wall-clock of the reference's `infer.infer` entry point on image FILES (decode + resize on the host workers, uint8
upload, table normalisation on the device, bf16 engine, fused post-processing, one gather, JSON) next to what bench.py
measures on device-resident tensors.  Writes N synthetic 1280x800 JPEGs to a scratch directory first."""
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

torch.backends.cudnn.benchmark = True
from odtk import infer
from odtk.model import Model

N = int(os.environ.get('PROBE_IMAGES', 192))
BATCH, WORKERS = 8, int(os.environ.get('PROBE_WORKERS', 16))
scratch = tempfile.mkdtemp(prefix='odtk_probe_')
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:800, 0:1280]
images = []
for k in range(N):
    base = np.stack([(xx + 3 * k) % 256, (yy * 2 + k) % 256, ((xx + yy) // 2) % 256], 2).astype(np.uint8)
    base[100:300, 200 + k:500 + k] = rng.integers(0, 255, 3)
    Image.fromarray(base, 'RGB').save(os.path.join(scratch, 'im%04d.jpg' % k), quality=90)
    images.append({'id': k, 'file_name': 'im%04d.jpg' % k, 'width': 1280, 'height': 800})
ann = os.path.join(scratch, 'ann.json')
json.dump({'images': images}, open(ann, 'w'))

sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)
model = Model(os.environ.get('PROBE_BACKBONE', 'ResNet50FPN'))
model.initialize(None)
for attempt in ('warm-up (MIOpen find, engine fold)', 'timed'):
    sync()
    t0 = time.time()
    infer.infer(model, scratch, os.path.join(scratch, 'det.json'), 800, 1333, BATCH, annotations=ann, world=1,
                verbose=False, num_workers=WORKERS)
    sync()
    dt = time.time() - t0
    print('%s: %d images in %.2f s = %.1f img/s (batch %d, %d loader workers, padded to 896x1280)' % (attempt, N, dt, N / dt, BATCH, WORKERS))

# the loader alone
from odtk.data import DataIterator
it = DataIterator(scratch, 800, 1333, BATCH, 128, 1, ann, training=False, num_workers=WORKERS)
t0 = time.time()
n = 0
for data, ids, ratios in it:
    n += data.shape[0]
sync()
dt = time.time() - t0
print('loader alone (decode + resize + collate + upload + normalise): %d images in %.2f s = %.1f img/s' % (n, dt, n / dt))
