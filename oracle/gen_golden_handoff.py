#!/usr/bin/env python
"""Golden fixtures for the detection hand-off: the REFERENCE's OWN per-detection loop, run here.

TEST INFRASTRUCTURE ONLY (build container: needs /root/reference).  The loop lives inside `infer()` of the reference's
odtk/infer.py (:104-148), a module that cannot be imported here (apex, pycocotools, DALI at import time).  So -- like
tests/test_cli.py does for `parse()` -- the statements are lifted out of the function with `ast` at generation time: the
`if is_master:` block of `infer()` from `results = [r.cpu() for r in results]` through the `for scores, boxes, ...` loop
(everything up to the file writing), compiled as they stand and executed on seeded inputs with the names they use bound to
plain objects: `results`, `rotated_bbox`, `data_iterator` (a stub with `.coco.dataset` / `.coco.getCatIds()`), `np`, `torch`
and the reference's own `rotate_box` (odtk/utils.py:83-101, lifted the same way).  Nothing of the reference is copied into
the repository: only the inputs (npz) and the detections it produced (JSON text, doubles round-trip exactly) are written to
tests/golden/handoff_*.{npz,json}.

    python oracle/gen_golden_handoff.py
"""
import ast
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = '/root/reference/odtk'
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def fake_results(n, d, rotated, seed):
    """Seeded stand-in for the five gathered result tensors (scores [N, D] NMS order with zero padding rows, boxes, classes,
    image ids with a DistributedSampler padding duplicate, resize ratios)."""
    g = torch.Generator().manual_seed(seed)
    scores = torch.rand(n, d, generator=g).sort(1, descending=True)[0]
    scores[:, d // 2:] *= (torch.rand(n, d - d // 2, generator=g) > 0.5)
    xy = torch.rand(n, d, 2, generator=g) * 900
    wh = torch.rand(n, d, 2, generator=g) * 300
    boxes = torch.cat([xy, xy + wh], 2)
    if rotated:
        th = (torch.rand(n, d, generator=g) - 0.5) * 3
        boxes = torch.cat([boxes, th.sin()[..., None], th.cos()[..., None]], 2)
    classes = torch.randint(0, 80, (n, d), generator=g).float()
    ids = torch.randint(0, 50_000_000, (n,), generator=g)
    ids[n - 1] = ids[0]
    ratios = torch.rand(n, generator=g) + 0.5
    return scores, boxes, classes, ids, ratios


def _function(path, name):
    tree = ast.parse(open(path).read())
    return next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)


def reference_loop():
    """-> callable(results, rotated_bbox, category_ids or None) -> list of detection dicts, running the reference's statements."""
    infer_fn = _function(os.path.join(REFERENCE, 'infer.py'), 'infer')
    master = next(n for n in infer_fn.body if isinstance(n, ast.If) and isinstance(n.test, ast.Name) and n.test.id == 'is_master')
    body, stop = [], False
    for stmt in master.body:
        body.append(stmt)
        if isinstance(stmt, ast.For):                       # the per-image / per-detection loop is the last statement wanted
            stop = True
            break
    assert stop and isinstance(body[0], ast.Assign), 'reference infer.py no longer has the shape this lifter expects'
    code = compile(ast.Module(body=body, type_ignores=[]), 'reference odtk/infer.py (if is_master: ... for ... loop)', 'exec')
    scope_utils = {'np': np}
    rotate_fn = _function(os.path.join(REFERENCE, 'utils.py'), 'rotate_box')
    exec(compile(ast.Module(body=[rotate_fn], type_ignores=[]), 'reference odtk/utils.py rotate_box', 'exec'), scope_utils)

    class _Coco:
        def __init__(self, category_ids):
            self.dataset = {'images': []}
            if category_ids is not None:
                self.dataset['annotations'] = []
            self._ids = category_ids

        def getCatIds(self):
            return self._ids

    class _Iterator:
        def __init__(self, category_ids):
            self.coco = _Coco(category_ids)

    def run(results, rotated_bbox, category_ids=None):
        scope = {'results': [r.clone() for r in results], 'rotated_bbox': rotated_bbox, 'data_iterator': _Iterator(category_ids),
                 'np': np, 'torch': torch, 'rotate_box': scope_utils['rotate_box']}
        exec(code, scope)
        return scope['detections']
    return run


CASES = [('handoff_axis', 9, 12, False, 0, None), ('handoff_axis_categories', 6, 20, False, 3, list(range(100, 180))),
         ('handoff_rotated', 7, 10, True, 1, None), ('handoff_rotated_categories', 5, 16, True, 4, list(range(1, 161, 2)))]


def main():
    run = reference_loop()
    for name, n, d, rotated, seed, cats in CASES:
        results = fake_results(n, d, rotated, seed)
        dets = run(results, rotated, cats)
        np.savez(os.path.join(GOLDEN, name + '.npz'), scores=results[0].numpy(), boxes=results[1].numpy(), classes=results[2].numpy(),
                 ids=results[3].numpy(), ratios=results[4].numpy(), rotated=np.array(rotated),
                 category_ids=np.array(cats if cats is not None else [], dtype=np.int64))
        for det in dets:                                     # numpy scalars (theta) -> python floats, same value
            det['bbox'] = [float(v) for v in det['bbox']]
        with open(os.path.join(GOLDEN, name + '.json'), 'w') as f:
            json.dump(dets, f)
        print('%s: %d detections from %d images' % (name, len(dets), n))


if __name__ == '__main__':
    sys.exit(main())
