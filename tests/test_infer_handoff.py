"""Detection hand-off (odtk/infer.py): the vectorised COCO conversion against
  (1) fixtures produced by the REFERENCE's OWN per-detection loop (odtk/infer.py:104-148 + utils.py:83-101, lifted out of
      its module and executed by oracle/gen_golden_handoff.py; tests/golden/handoff_*.{npz,json}), re-generated live when
      /root/reference is present,
  (2) a per-detection restatement of that loop on further seeded inputs,
and the end-to-end `infer` driver on a stub model (CPU)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from odtk import infer, parallel

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _same_detections(got, ref, rotated):
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert (g['image_id'], g['score'], g['category_id']) == (r['image_id'], r['score'], r['category_id'])
        assert g['bbox'][:4] == r['bbox'][:4]                         # doubles, bit for bit
        if rotated:
            assert abs(g['bbox'][4] - r['bbox'][4]) <= 1e-15
            assert np.allclose(g['segmentation'][0], r['segmentation'][0], rtol=0, atol=1e-9)   # matmul order / FMA
            assert set(g) == set(r) == {'image_id', 'score', 'category_id', 'bbox', 'segmentation'}
        else:
            assert set(g) == set(r) == {'image_id', 'score', 'category_id', 'bbox'}


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'handoff_*.npz'))), ids=os.path.basename)
def test_coco_conversion_equals_the_references_own_loop(path):
    """Fixtures = output of the reference's loop itself (not of a restatement)."""
    z = np.load(path)
    ref = json.load(open(path[:-4] + '.json'))
    rotated = bool(z['rotated'])
    cats = z['category_ids'].tolist() or None
    args = [torch.from_numpy(z[k]) for k in ('scores', 'boxes', 'classes', 'ids', 'ratios')]
    got = infer.detections_to_coco(*args, rotated_bbox=rotated, category_ids=cats)
    assert len(ref) > 40
    _same_detections(got, ref, rotated)
    if not rotated:
        assert json.dumps(got) == json.dumps(ref)                     # ... and so is the file the reference would write


@pytest.mark.skipif(not os.path.isdir('/root/reference/odtk'), reason='reference tree not present (GPU box)')
@pytest.mark.parametrize('rotated,seed,cats', [(False, 11, None), (True, 12, None), (False, 13, list(range(7, 87))), (True, 14, list(range(80)))])
def test_coco_conversion_equals_the_reference_loop_run_live(rotated, seed, cats):
    from oracle import gen_golden_handoff as G
    run = G.reference_loop()
    args = G.fake_results(8, 25, rotated, seed)
    ref = run(args, rotated, cats)
    for det in ref:
        det['bbox'] = [float(v) for v in det['bbox']]
    got = infer.detections_to_coco(*args, rotated_bbox=rotated, category_ids=cats)
    _same_detections(got, ref, rotated)


def _reference_loop(scores, boxes, classes, ids, ratios, rotated):
    """Slow checker: one detection at a time, Python arithmetic on .tolist() values, as the reference."""
    out, seen = [], set()
    for s, b, c, image_id, ratio in zip(scores, boxes, classes, ids, ratios):
        image_id = image_id.item()
        if image_id in seen:
            continue
        seen.add(image_id)
        keep = (s > 0).nonzero(as_tuple=False)
        s = s[keep].view(-1)
        if rotated:
            b = b[keep, :].view(-1, 6).clone()
            b[:, :4] /= ratio
        else:
            b = b[keep, :].view(-1, 4) / ratio
        c = c[keep].view(-1).int()
        for score, box, cat in zip(s, b, c):
            det = {'image_id': image_id, 'score': score.item(), 'category_id': cat.item()}
            if rotated:
                x1, y1, x2, y2, sin, cos = box.tolist()
                theta = np.arctan2(sin, cos)
                w, h = x2 - x1 + 1, y2 - y1 + 1
                corners = np.stack([(x1, y1), (x1, y1 + h - 1), (x1 + w - 1, y1 + h - 1), (x1 + w - 1, y1)])
                cents = np.array([x1 + (w - 1) / 2, y1 + (h - 1) / 2])
                rot = np.vstack([np.stack([np.cos(theta), -np.sin(theta)]), np.stack([np.sin(theta), np.cos(theta)])])
                seg = (np.matmul(rot, (corners - cents).transpose(1, 0)).transpose(1, 0) + cents).reshape(-1).tolist()
                det['bbox'] = [x1, y1, w, h, theta]
                det['segmentation'] = [seg]
            else:
                x1, y1, x2, y2 = box.tolist()
                det['bbox'] = [x1, y1, x2 - x1 + 1, y2 - y1 + 1]
            out.append(det)
    return out


def _fake_results(n, d, rotated, seed):
    g = torch.Generator().manual_seed(seed)
    scores = torch.rand(n, d, generator=g).sort(1, descending=True)[0]
    scores[:, d // 2:] *= (torch.rand(n, d - d // 2, generator=g) > 0.5)       # zero-score padding rows
    xy = torch.rand(n, d, 2, generator=g) * 900
    wh = torch.rand(n, d, 2, generator=g) * 300
    boxes = torch.cat([xy, xy + wh], 2)
    if rotated:
        th = (torch.rand(n, d, generator=g) - 0.5) * 3
        boxes = torch.cat([boxes, th.sin()[..., None], th.cos()[..., None]], 2)
    classes = torch.randint(0, 80, (n, d), generator=g).float()
    ids = torch.randint(0, 50_000_000, (n,), generator=g)
    ids[n - 1] = ids[0]                                                          # DistributedSampler padding duplicate
    ratios = torch.rand(n, generator=g) + 0.5
    return scores, boxes, classes, ids, ratios


def test_coco_conversion_matches_the_reference_loop_axis_aligned():
    args = _fake_results(9, 12, False, 0)
    got = infer.detections_to_coco(*args)
    ref = _reference_loop(*args, rotated=False)
    assert got == ref                                    # ids, scores, categories, bbox doubles: identical
    assert json.dumps(got) == json.dumps(ref)            # ... and so is the file the reference would write


def test_coco_conversion_matches_the_reference_loop_rotated():
    args = _fake_results(7, 10, True, 1)
    got = infer.detections_to_coco(*args, rotated_bbox=True)
    ref = _reference_loop(*args, rotated=True)
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert (g['image_id'], g['score'], g['category_id']) == (r['image_id'], r['score'], r['category_id'])
        assert g['bbox'][:4] == r['bbox'][:4]
        assert abs(g['bbox'][4] - r['bbox'][4]) <= 1e-15
        assert np.allclose(g['segmentation'][0], r['segmentation'][0], rtol=0, atol=1e-9)   # matmul order / FMA


def test_category_map_empty_and_large_ids():
    s, b, c, i, r = _fake_results(3, 4, False, 2)
    cats = list(range(100, 180))
    got = infer.detections_to_coco(s, b, c, i, r, category_ids=cats)
    assert all(d['category_id'] == cats[int(k)] for d, k in zip(got, c[:2][s[:2] > 0]))
    assert infer.detections_to_coco(torch.zeros(2, 5), torch.zeros(2, 5, 4), torch.zeros(2, 5), torch.tensor([1, 2]),
                                    torch.ones(2)) == []
    # ids above 2^24 survive the packed float tensor bit for bit
    big = torch.tensor([2 ** 31 - 1, 16777217, 5])
    packed = parallel.pack_detections(s, b, c, big, r)
    assert parallel.unpack_detections(packed, 4, 4)[3].tolist() == big.tolist()


class _StubModel:
    """Returns canned detections keyed by the mean of the image (deterministic, no GPU)."""
    def __init__(self, d):
        self.d = d

    def __call__(self, images):
        n = images.shape[0]
        base = images.mean((1, 2, 3)).view(n, 1)
        scores = (torch.arange(self.d, 0, -1).float().view(1, -1) / self.d + base).clamp(min=0)
        scores[:, -1] = 0
        boxes = torch.arange(n * self.d * 4).float().view(n, self.d, 4)
        boxes[:, :, 2:] += boxes[:, :, :2]
        return scores, boxes, torch.full((n, self.d), 3.0)


def test_infer_driver_writes_the_reference_document(tmp_path):
    batches = [(torch.full((2, 3, 8, 8), 0.1 * k), torch.tensor([10 * k, 10 * k + 1]), torch.tensor([1.0, 2.0])) for k in range(3)]
    out = tmp_path / 'det.json'
    dataset = {'images': [{'id': 0}], 'categories': [{'id': 3}]}
    dets = infer.infer_batches(_StubModel(5), batches, detections_file=str(out), dataset=dataset)
    assert len(dets) == 6 * 4                           # 6 images x (5 - 1 zero-score row)
    doc = json.load(open(out))
    assert doc['annotations'] == dets and doc['images'] == dataset['images'] and doc['categories'] == dataset['categories']
    assert [d['image_id'] for d in dets[:4]] == [0] * 4 and dets[0]['score'] >= dets[1]['score']
    # ratio 2.0 halves the second image's boxes
    second = [d for d in dets if d['image_id'] == 1][0]
    assert second['bbox'][0] == float(torch.tensor(20.0) / 2.0)
