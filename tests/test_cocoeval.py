"""odtk/cocoeval.py (restatement of pycocotools' COCOeval, 'bbox'): hand-computed cases + the loop-by-loop
restatement in oracle/cocoeval_loops.py on random data."""
import io
import random

import numpy as np
import pytest

from odtk.cocoeval import COCOeval, box_iou, polygon_iou
from odtk.data import CocoIndex
from oracle import cocoeval_loops


def _gt(boxes, image_id=1, cat=1, crowd=(), first_id=1, areas=None):
    return [{'id': first_id + k, 'image_id': image_id, 'category_id': cat, 'bbox': list(map(float, b)),
             'area': float(b[2] * b[3]) if areas is None else areas[k], 'iscrowd': int(k in crowd)}
            for k, b in enumerate(boxes)]


def _dt(rows, image_id=1, cat=1):
    return [{'image_id': image_id, 'category_id': cat, 'score': s, 'bbox': list(map(float, b))} for s, b in rows]


def _stats(gt, dt, images=(1,), cats=(1,)):
    index = CocoIndex(dataset={'images': [{'id': i} for i in images], 'annotations': gt,
                               'categories': [{'id': c} for c in cats]})
    ev = COCOeval(index, index.loadRes(dt), 'bbox')
    ev.evaluate()
    ev.accumulate()
    text = io.StringIO()
    stats = ev.summarize(out=lambda line: text.write(line + '\n'))
    return stats, ev, text.getvalue()


def test_perfect_detections():
    gt = _gt([[0, 0, 10, 10], [50, 50, 40, 40]])
    stats, _, text = _stats(gt, _dt([(0.9, [0, 0, 10, 10]), (0.8, [50, 50, 40, 40])]))
    one = pytest.approx(1.0, abs=1e-12)                                       # tp / (tp + fp + eps)
    assert stats[0] == one and stats[1] == one and stats[2] == one
    assert stats[3] == one and stats[4] == one and stats[5] == -1.0          # small, medium; no large box
    assert stats[6] == 0.5 and stats[7] == stats[8] == 1.0                    # AR@1 sees one of the two boxes
    lines = text.splitlines()
    assert lines[0] == ' Average Precision  (AP) @[ IoU=0.50:0.95 | area=   all | maxDets=100 ] = 1.000'
    assert lines[6] == ' Average Recall     (AR) @[ IoU=0.50:0.95 | area=   all | maxDets=  1 ] = 0.500'
    assert lines[1].startswith(' Average Precision  (AP) @[ IoU=0.50      | area=   all')


def test_a_false_positive_ahead_of_two_hits_gives_two_thirds():
    gt = _gt([[0, 0, 10, 10], [50, 50, 40, 40]])
    dt = _dt([(0.9, [200, 200, 10, 10]), (0.8, [0, 0, 10, 10]), (0.7, [50, 50, 40, 40])])
    stats, _, _ = _stats(gt, dt)
    # TP 0,1,2 / FP 1,1,1 -> precision 0, 1/2, 2/3, envelope 2/3 at every recall level, every IoU threshold
    assert stats[0] == pytest.approx(2 / 3, abs=1e-12) and stats[1] == pytest.approx(2 / 3, abs=1e-12)
    assert stats[8] == 1.0 and stats[6] == 0.0                                # the top detection is the miss


def test_iou_077_counts_at_six_of_ten_thresholds():
    stats, _, _ = _stats(_gt([[0, 0, 10, 10]]), _dt([(0.9, [0, 0, 10, 7.7])]))
    assert box_iou([[0, 0, 10, 7.7]], [[0, 0, 10, 10]], [False])[0, 0] == pytest.approx(0.77)
    assert stats[0] == pytest.approx(0.6) and stats[1] == pytest.approx(1.0) and stats[2] == pytest.approx(1.0)
    assert stats[3] == pytest.approx(0.6) and stats[4] == -1.0 and stats[5] == -1.0
    assert stats[8] == pytest.approx(0.6)


def test_crowd_box_absorbs_detections_without_penalty():
    gt = _gt([[0, 0, 100, 100], [0, 0, 10, 10]], crowd=(0,))
    dt = _dt([(0.9, [0, 0, 10, 10]), (0.8, [50, 50, 10, 10]), (0.7, [60, 60, 10, 10])])
    stats, ev, _ = _stats(gt, dt)
    assert stats[0] == pytest.approx(1.0) and stats[8] == 1.0
    first = ev.evalImgs[0]
    assert first['dtMatches'][0].tolist() == [2, 1, 1]                        # the crowd box is taken twice
    assert first['dtIgnore'][0].tolist() == [False, True, True]
    assert box_iou([[50, 50, 10, 10]], [[0, 0, 100, 100]], [True])[0, 0] == 1.0


def test_equal_iou_goes_to_the_later_box_and_area_ranges_ignore_outsiders():
    gt = _gt([[0, 0, 10, 10], [0, 0, 10, 10]])
    _, ev, _ = _stats(gt, _dt([(0.9, [0, 0, 10, 10])]))
    assert ev.evalImgs[0]['dtMatches'][:, 0].tolist() == [2] * 10
    gt = _gt([[0, 0, 10, 10]])
    stats, _, _ = _stats(gt, _dt([(0.9, [300, 300, 200, 200]), (0.8, [0, 0, 10, 10])]))
    assert stats[0] == pytest.approx(0.5) and stats[3] == pytest.approx(1.0) and stats[5] == -1.0


def test_no_detections_and_annotation_area_field():
    stats, _, _ = _stats(_gt([[0, 0, 10, 10]]), [])
    assert stats[0] == 0.0 and stats[8] == 0.0 and stats[4] == -1.0
    gt = _gt([[0, 0, 10, 10]], areas=[5000.0])                                # the file's `area`, not w * h, picks the range
    stats, _, _ = _stats(gt, _dt([(0.9, [0, 0, 10, 10])]))
    assert stats[3] == -1.0 and stats[4] == pytest.approx(1.0)
    with pytest.raises(NotImplementedError):
        index = CocoIndex(dataset={'images': [], 'annotations': []})
        COCOeval(index, index, 'keypoints')


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_equals_the_loop_restatement_on_random_scenes(seed):
    rng = random.Random(seed)
    images, cats = [3, 5, 8, 13], [2, 9, 4]
    gt, dt = [], []
    for img in images:
        for _ in range(rng.randint(0, 9)):
            w, h = rng.choice([6, 20, 50, 120]), rng.choice([6, 20, 50, 120])
            box = [rng.randint(0, 200), rng.randint(0, 200), w, h]
            gt.append({'id': len(gt) + 1, 'image_id': img, 'category_id': rng.choice(cats), 'bbox': [float(v) for v in box],
                       'area': float(w * h), 'iscrowd': int(rng.random() < 0.15)})
            for _ in range(rng.randint(0, 3)):                                # detections near this box, sometimes exact copies
                jitter = [rng.choice([0, 0, 1, -2, 5, 9]) for _ in range(4)]
                cand = [float(box[0] + jitter[0]), float(box[1] + jitter[1]), float(max(1, w + jitter[2])), float(max(1, h + jitter[3]))]
                dt.append({'image_id': img, 'category_id': gt[-1]['category_id'] if rng.random() < 0.8 else rng.choice(cats),
                           'score': rng.choice([0.9, 0.8, 0.8, round(rng.random(), 2)]), 'bbox': cand})
        for _ in range(rng.randint(0, 4)):                                    # strays
            dt.append({'image_id': img, 'category_id': rng.choice(cats), 'score': round(rng.random(), 2),
                       'bbox': [float(rng.randint(0, 300)), float(rng.randint(0, 300)), float(rng.randint(4, 150)), float(rng.randint(4, 150))]})
    index = CocoIndex(dataset={'images': [{'id': i} for i in images], 'annotations': gt, 'categories': [{'id': c} for c in cats]})
    res = index.loadRes(dt)
    ev = COCOeval(index, res, 'bbox')
    ev.evaluate()
    ev.accumulate()
    got = ev.summarize(out=lambda line: None)
    want = cocoeval_loops.stats(gt, list(res.anns.values()), images, cats)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    assert 0 < got[0] < 1


# ---- iouType 'segm': rotated boxes as polygons (reference infer.py:166), exact polygon IoU ----------------------------
def _quad(x, y, w, h, theta=0.0):
    """Corners of a w x h rectangle whose top-left corner sits at (x, y) before it is turned by theta about its centre."""
    cx, cy = x + w / 2, y + h / 2
    c, s = np.cos(theta), np.sin(theta)
    pts = [(-w / 2, -h / 2), (w / 2, -h / 2), (w / 2, h / 2), (-w / 2, h / 2)]
    return [[float(v) for px, py in pts for v in (cx + c * px - s * py, cy + s * px + c * py)]]


def test_polygon_iou_of_axis_aligned_rectangles_is_the_box_iou():
    rng = np.random.default_rng(0)
    boxes = np.concatenate([rng.random((40, 2)) * 60, rng.random((40, 2)) * 50 + 2], 1)
    anns = [{'id': i + 1, 'segmentation': _quad(*b)} for i, b in enumerate(boxes)]
    crowd = rng.random(20) < 0.3
    got = polygon_iou(anns[:20], anns[20:], crowd)
    want = box_iou(boxes[:20], boxes[20:], crowd)
    assert np.abs(got - want).max() < 1e-12 and (want > 0).sum() > 50
    # clockwise corner order and a different starting corner describe the same region
    flipped = [{'id': a['id'], 'segmentation': [list(np.asarray(a['segmentation'][0]).reshape(-1, 2)[::-1].ravel())]} for a in anns[:20]]
    assert np.abs(polygon_iou(flipped, anns[20:], crowd) - want).max() < 1e-12


def test_polygon_iou_known_rotations():
    sq = {'id': 1, 'segmentation': _quad(0, 0, 10, 10)}
    # a square turned by 45 degrees about the same centre: the intersection is a regular octagon, IoU = 1 / sqrt(2)... of the union:
    # inter = 200 (sqrt(2) - 1), union = 200 - inter
    turned = {'id': 2, 'segmentation': _quad(0, 0, 10, 10, np.pi / 4)}
    inter = 200 * (np.sqrt(2) - 1)
    assert polygon_iou([turned], [sq], [False])[0, 0] == pytest.approx(inter / (200 - inter), abs=1e-12)
    assert polygon_iou([turned], [sq], [True])[0, 0] == pytest.approx(inter / 100, abs=1e-12)       # crowd: over the detection
    # a 20 x 2 bar turned by 90 degrees about the centre of a 20 x 2 bar: they share a 2 x 2 square
    bar, cross = {'id': 3, 'segmentation': _quad(0, 9, 20, 2)}, {'id': 4, 'segmentation': _quad(0, 9, 20, 2, np.pi / 2)}
    assert polygon_iou([bar], [cross], [False])[0, 0] == pytest.approx(4 / 76, abs=1e-12)
    far = {'id': 5, 'segmentation': _quad(100, 100, 5, 5, 0.3)}
    assert polygon_iou([far, sq], [sq], [False]).tolist() == [[0.0], [pytest.approx(1.0, abs=1e-12)]]
    touching = {'id': 6, 'segmentation': _quad(10, 0, 10, 10)}                                       # shares an edge only
    assert polygon_iou([touching], [sq], [False])[0, 0] == 0.0


def test_polygon_iou_refuses_what_it_cannot_do_exactly():
    with pytest.raises(NotImplementedError, match='not ONE polygon'):
        polygon_iou([{'id': 1, 'segmentation': {'counts': 'abc', 'size': [4, 4]}}], [{'id': 2, 'segmentation': _quad(0, 0, 1, 1)}], [False])
    with pytest.raises(NotImplementedError, match='not ONE polygon'):
        polygon_iou([{'id': 1, 'segmentation': _quad(0, 0, 1, 1) * 2}], [{'id': 2, 'segmentation': _quad(0, 0, 1, 1)}], [False])
    arrow = [[0, 0, 10, 0, 10, 10, 5, 2, 0, 10]]                                                     # a notch: not convex
    with pytest.raises(NotImplementedError, match='not convex'):
        polygon_iou([{'id': 1, 'segmentation': arrow}], [{'id': 2, 'segmentation': _quad(0, 0, 1, 1)}], [False])


def test_segm_evaluation_of_rotated_detections():
    """Rotated ground truth (RotatedCocoDataset form: x, y, w, h, theta + the polygon) against rotated detections: a
    detection with the right extent but the wrong angle is a miss under 'segm' and a hit under 'bbox'."""
    def ann(i, x, y, w, h, theta, **extra):
        return dict({'id': i, 'image_id': 1, 'category_id': 1, 'bbox': [x, y, w, h, theta], 'area': float(w * h), 'iscrowd': 0,
                     'segmentation': _quad(x, y, w, h, theta)}, **extra)
    gt = [ann(1, 10, 10, 60, 12, 0.5), ann(2, 100, 40, 30, 30, 0.0)]
    index = CocoIndex(dataset={'images': [{'id': 1}], 'annotations': gt, 'categories': [{'id': 1}]})

    def run(dets, kind):
        ev = COCOeval(index, index.loadRes(dets), kind)
        ev.evaluate()
        ev.accumulate()
        return ev.summarize(out=lambda line: None)

    def det(score, x, y, w, h, theta):
        return {'image_id': 1, 'category_id': 1, 'score': score, 'bbox': [x, y, w, h, theta], 'segmentation': _quad(x, y, w, h, theta)}
    right = [det(0.9, 10, 10, 60, 12, 0.5), det(0.8, 100, 40, 30, 30, 0.0)]
    assert run(right, 'segm')[0] == pytest.approx(1.0, abs=1e-12)
    wrong_angle = [det(0.9, 10, 10, 60, 12, -0.5), det(0.8, 100, 40, 30, 30, 0.0)]
    assert run(wrong_angle, 'bbox')[0] == pytest.approx(1.0, abs=1e-12)       # the axis-aligned fields agree ...
    s = run(wrong_angle, 'segm')
    assert s[8] == pytest.approx(0.5) and s[0] < 0.6                          # ... the regions do not: one of two found
    slightly_off = [det(0.9, 10, 10, 60, 12, 0.52), det(0.8, 101, 40, 30, 30, 0.0)]
    s = run(slightly_off, 'segm')
    assert 0.5 < s[0] < 1.0 and s[1] == pytest.approx(1.0, abs=1e-12)         # hits at IoU 0.5, not at 0.95


def test_segm_on_the_rotated_annotation_fixture():
    """The rotated data-set fixture (bbox = x, y, w, h, theta; no polygon in the file: derived with infer.rotated_corners):
    its own boxes as detections score AP 1, the same boxes with the angles negated do not."""
    import json
    import os
    from odtk import infer
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data', 'annotations_rotated.json')
    index = CocoIndex(path)
    gt = json.load(open(path))['annotations']

    def dets(sign):
        out = []
        for k, a in enumerate(gt):
            x, y, w, h, theta = (list(a['bbox']) + [0.0])[:5]                  # plain boxes: theta = 0, as RotatedCocoDataset reads them
            seg = infer.rotated_corners([x], [y], [w], [h], [sign * theta])[0].tolist()
            out.append({'image_id': a['image_id'], 'category_id': a['category_id'], 'score': 0.9 - 0.01 * k,
                        'bbox': [x, y, w, h, sign * theta], 'segmentation': [seg]})
        return out

    def run(d):
        ev = COCOeval(index, index.loadRes(d), 'segm')
        ev.evaluate()
        ev.accumulate()
        return ev.summarize(out=lambda line: None)
    assert run(dets(1.0))[0] == pytest.approx(1.0, abs=1e-12)
    assert any(len(a['bbox']) == 5 and abs(a['bbox'][4]) > 0.3 for a in gt)
    assert run(dets(-1.0))[0] < 0.9


def test_polygon_iou_against_pixel_counting():
    """The docstring's claim: the exact polygon IoU and an IoU counted on rasterised masks (what pycocotools does; here pixel
    centres inside the polygon) differ by boundary pixels only -- under 0.03 for boxes of a few hundred pixels and more."""
    rng = np.random.default_rng(4)
    ys, xs = np.mgrid[0:160, 0:160]
    px, py = xs.ravel() + 0.5, ys.ravel() + 0.5

    def mask(seg):
        pts = np.asarray(seg[0]).reshape(-1, 2)
        inside = np.ones(px.shape, bool)
        sign = np.sign(np.sum(pts[:, 0] * np.roll(pts[:, 1], -1) - np.roll(pts[:, 0], -1) * pts[:, 1]))
        for (ax, ay), (bx, by) in zip(pts, np.roll(pts, -1, axis=0)):
            inside &= sign * ((bx - ax) * (py - ay) - (by - ay) * (px - ax)) >= 0
        return inside
    worst, compared = 0.0, 0
    for _ in range(60):
        x, y = rng.random(2) * 60 + 30
        w, h = rng.random(2) * 50 + 20
        a = _quad(x, y, w, h, rng.uniform(-1.5, 1.5))
        b = _quad(x + rng.uniform(-12, 12), y + rng.uniform(-12, 12), w * rng.uniform(0.7, 1.3), h * rng.uniform(0.7, 1.3), rng.uniform(-1.5, 1.5))
        exact = polygon_iou([{'id': 1, 'segmentation': a}], [{'id': 2, 'segmentation': b}], [False])[0, 0]
        ma, mb = mask(a), mask(b)
        counted = (ma & mb).sum() / max((ma | mb).sum(), 1)
        worst = max(worst, abs(exact - counted))
        compared += exact > 0.2
    assert worst < 0.03 and compared > 30, (worst, compared)
