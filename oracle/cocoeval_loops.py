"""TEST INFRASTRUCTURE ONLY: the COCO bbox evaluation written as plain nested loops, one detection, one
ground-truth box and one IoU threshold at a time -- the published algorithm of pycocotools' `COCOeval`
(nvidia/cocoapi master; package absent from this image, so this restatement is pinned to hand-computed cases
only: "parity unpinned" against the package itself).  tests/test_cocoeval.py checks the product's vectorised
`odtk/cocoeval.py` against it on random data.  Returns the twelve `stats`.
"""
import numpy as np

IOU_THRS = np.linspace(.5, 0.95, 10)
REC_THRS = np.linspace(.0, 1.00, 101)
MAX_DETS = [1, 10, 100]
AREAS = [[0, 1e10], [0, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e10]]


def iou_one(d, g, crowd):
    w = min(d[0] + d[2], g[0] + g[2]) - max(d[0], g[0])
    h = min(d[1] + d[3], g[1] + g[3]) - max(d[1], g[1])
    if w <= 0 or h <= 0:
        return 0.0
    inter = w * h
    return inter / (d[2] * d[3] if crowd else d[2] * d[3] + g[2] * g[3] - inter)


def evaluate_pair(gt, dt, rng, t):
    """One (image, category, area range, IoU threshold): per detection (score order) -> (matched gt id, ignored)."""
    gt = sorted(gt, key=lambda g: bool(g.get('iscrowd', 0)) or g['area'] < rng[0] or g['area'] > rng[1])   # stable
    g_ignore = [bool(g.get('iscrowd', 0)) or g['area'] < rng[0] or g['area'] > rng[1] for g in gt]
    dt = sorted(dt, key=lambda d: -d['score'])[:MAX_DETS[-1]]                                               # stable
    taken = [0] * len(gt)
    out = []
    for d in dt:
        best, m = min(t, 1 - 1e-10), -1
        for gi, g in enumerate(gt):
            if taken[gi] > 0 and not g.get('iscrowd', 0):
                continue
            if m > -1 and not g_ignore[m] and g_ignore[gi]:
                break
            v = iou_one(d['bbox'], g['bbox'], bool(g.get('iscrowd', 0)))
            if v < best:
                continue
            best, m = v, gi
        if m == -1:
            out.append((0, d['area'] < rng[0] or d['area'] > rng[1], d['score']))
        else:
            taken[m] = d['id']
            out.append((gt[m]['id'], g_ignore[m], d['score']))
    return out, g_ignore


def stats(gt_anns, dt_anns, img_ids, cat_ids):
    img_ids, cat_ids = sorted(set(img_ids)), sorted(set(cat_ids))
    T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(cat_ids), len(AREAS), len(MAX_DETS)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    for k, cat in enumerate(cat_ids):
        for a, rng in enumerate(AREAS):
            for ti, t in enumerate(IOU_THRS):
                per_image = []
                for img in img_ids:
                    gt = [g for g in gt_anns if g['image_id'] == img and g['category_id'] == cat]
                    dt = [d for d in dt_anns if d['image_id'] == img and d['category_id'] == cat]
                    if gt or dt:
                        per_image.append(evaluate_pair(gt, dt, rng, t))
                if not per_image:
                    continue
                n_regular = sum(1 for _, gi in per_image for v in gi if not v)
                if n_regular == 0:
                    continue
                for m, cap in enumerate(MAX_DETS):
                    rows = [r for dets, _ in per_image for r in dets[:cap]]
                    rows.sort(key=lambda r: -r[2])                                                          # stable
                    tp = fp = 0
                    rc, pr = [], []
                    for match, ignored, _ in rows:
                        if not ignored:
                            tp += 1 if match != 0 else 0
                            fp += 1 if match == 0 else 0
                        rc.append(tp / n_regular)
                        pr.append(tp / (fp + tp + np.spacing(1)))
                    recall[ti, k, a, m] = rc[-1] if rows else 0
                    for i in range(len(pr) - 1, 0, -1):
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    q = [0.0] * R
                    for ri, r in enumerate(REC_THRS):
                        pi = int(np.searchsorted(rc, r, side='left')) if rc else 0
                        if pi >= len(pr):
                            break
                        q[ri] = pr[pi]
                    precision[ti, :, k, a, m] = q

    def mean(table):
        valid = table[table > -1]
        return float(valid.mean()) if valid.size else -1.0

    return np.array([
        mean(precision[:, :, :, 0, 2]), mean(precision[0:1, :, :, 0, 2]), mean(precision[5:6, :, :, 0, 2]),
        mean(precision[:, :, :, 1, 2]), mean(precision[:, :, :, 2, 2]), mean(precision[:, :, :, 3, 2]),
        mean(recall[:, :, 0, 0]), mean(recall[:, :, 0, 1]), mean(recall[:, :, 0, 2]),
        mean(recall[:, :, 1, 2]), mean(recall[:, :, 2, 2]), mean(recall[:, :, 3, 2])])
