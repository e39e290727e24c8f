"""Generate tests/golden/data/ by running the REFERENCE's own odtk/data.py (build container only).

    python -m oracle.gen_golden_data          # from the repo root; needs /root/reference

TEST INFRASTRUCTURE ONLY.  Writes a five-image COCO-style data set (PNG files + annotations.json, drawn
here from a seeded generator) and `expected.npz` = what the unmodified reference `CocoDataset` /
`RotatedCocoDataset` returned for it: inference items (normalised pixels, id, ratio), one collated
inference batch, and seeded training items + a collated training batch (quarter turns and flips on,
colour jitter off).  The fixtures travel to the GPU box; /root/reference does not.

Shims needed to import reference data.py here: `pycocotools.coco.COCO` (absent; bound to the port's
`CocoIndex` -- index look-ups only, no arithmetic) and `torchvision.transforms.functional` (absent; the
four `adjust_*` names are never called with jitter off).
"""
import importlib
import json
import os
import random
import sys
import types

import numpy as np
import torch
from PIL import Image, ImageDraw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'retinanet-examples_amd'))

from oracle import ref_loader          # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'data')
SIZES = [(131, 97), (180, 240), (256, 256), (333, 250), (90, 310)]          # (width, height)
CATEGORY_IDS = [7, 3, 11]                                                    # file order != sorted order


def draw_dataset():
    rng = random.Random(20260924)
    os.makedirs(OUT, exist_ok=True)
    images, annotations = [], []
    for k, (w, h) in enumerate(SIZES):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 7) % 256], 2).astype(np.uint8)
        im = Image.fromarray(base, 'RGB')
        draw = ImageDraw.Draw(im)
        image_id = 100 + 3 * k
        for _ in range(0 if k == 2 else rng.randint(1, 4)):                   # image 2 has no annotation
            bw, bh = rng.randint(8, w // 2), rng.randint(8, h // 2)
            x, y = rng.randint(0, w - bw - 1), rng.randint(0, h - bh - 1)
            draw.rectangle([x, y, x + bw, y + bh], fill=tuple(rng.randint(0, 255) for _ in range(3)))
            annotations.append({'id': len(annotations) + 1, 'image_id': image_id, 'category_id': rng.choice(CATEGORY_IDS),
                                'bbox': [x + 0.5 * rng.randint(0, 1), float(y), float(bw), float(bh)],
                                'area': float(bw * bh), 'iscrowd': 0})
        name = 'im%d.png' % k
        im.save(os.path.join(OUT, name), optimize=True)
        images.append({'id': image_id, 'file_name': name, 'width': w, 'height': h})
    annotations.append({'id': len(annotations) + 1, 'image_id': images[0]['id'], 'category_id': 3,
                        'bbox': [5.0, 5.0, 0.5, 0.25], 'area': 0.125, 'iscrowd': 0})          # sub-pixel: skipped
    doc = {'images': images, 'annotations': annotations,
           'categories': [{'id': c, 'name': 'c%d' % c} for c in CATEGORY_IDS]}
    with open(os.path.join(OUT, 'annotations.json'), 'w') as f:
        json.dump(doc, f, indent=1)
    rotated = json.loads(json.dumps(doc))
    for k, ann in enumerate(rotated['annotations']):
        if k % 2 == 0:
            ann['bbox'] = ann['bbox'] + [round(rng.uniform(-0.7, 0.7), 3)]
    with open(os.path.join(OUT, 'annotations_rotated.json'), 'w') as f:
        json.dump(rotated, f, indent=1)


def reference_data():
    """The reference's odtk.data module, imported with the two absent third-party modules stubbed."""
    from odtk.data import CocoIndex                      # the port's index (see the module docstring)
    stubs = {}
    stubs['pycocotools'] = types.ModuleType('pycocotools')
    stubs['pycocotools.coco'] = types.ModuleType('pycocotools.coco')
    stubs['pycocotools.coco'].COCO = CocoIndex
    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    tvf = types.ModuleType('torchvision.transforms.functional')
    for name in ('adjust_brightness', 'adjust_contrast', 'adjust_hue', 'adjust_saturation'):
        setattr(tvf, name, None)
    stubs.update({'torchvision': tv, 'torchvision.transforms': tvt, 'torchvision.transforms.functional': tvf})
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        with ref_loader._reference_modules():
            return importlib.import_module('odtk.data')
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def main():
    draw_dataset()
    ref = reference_data()
    ann = os.path.join(OUT, 'annotations.json')
    out = {}

    ds = ref.CocoDataset(OUT, resize=128, max_size=200, stride=32, annotations=ann, training=False)
    items = [ds[i] for i in range(len(ds))]
    for i, (pix, image_id, ratio) in enumerate(items):
        out['infer_pixels_%d' % i] = pix.numpy()
        out['infer_id_%d' % i] = np.int64(image_id)
        out['infer_ratio_%d' % i] = np.float64(ratio)
    batch, ids, ratios = ds.collate_fn(items[:4])
    out.update(infer_batch=batch.numpy(), infer_batch_ids=ids.numpy(), infer_batch_ratios=ratios.numpy())

    for tag, cls, path, extra in (('train', ref.CocoDataset, ann, {}),
                                  ('rtrain', ref.RotatedCocoDataset, os.path.join(OUT, 'annotations_rotated.json'), {}),
                                  ('rabs', ref.RotatedCocoDataset, os.path.join(OUT, 'annotations_rotated.json'),
                                   {'absolute_angle': True})):
        ds = cls(OUT, resize=[96, 160], max_size=220, stride=32, annotations=path, training=True,
                 rotate_augment=True, **extra)
        random.seed(1234)
        items = [ds[i % len(ds)] for i in range(10)]
        for i, (pix, target) in enumerate(items):
            out['%s_pixels_%d' % (tag, i)] = pix.numpy()
            out['%s_target_%d' % (tag, i)] = target.numpy()
        batch, targets = ds.collate_fn(items[:5])
        out['%s_batch' % tag] = batch.numpy()
        out['%s_batch_targets' % tag] = targets.numpy()

    np.savez_compressed(os.path.join(OUT, 'expected.npz'), **out)
    size = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print('wrote %d arrays, fixture directory %.1f KB' % (len(out), size / 1024))


if __name__ == '__main__':
    main()
