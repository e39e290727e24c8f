"""COCO detection metrics (AP / AR over IoU 0.50:0.95) for the detections the path hands off -- the
acceptance metric of the north star ("COCO mAP within +-0.1 of the reference").

The reference calls `pycocotools.cocoeval.COCOeval(gt, dt, 'bbox')` (infer.py:160-172; nvidia/cocoapi master,
un-pinned, ABSENT from this image and from /root/reference).  This module restates that package's published
algorithm for iouType 'bbox' -- parity is pinned to hand-computed cases only (tests/test_cocoeval.py), not to
the package itself:

  evaluate    per (image, category): IoU of [x, y, w, h] boxes without the +1 pixel convention, union = the
              detection's area against a crowd box; detections in score order (stable), ground truth with the
              non-ignored boxes first; per IoU threshold a detection takes the best still-free ground truth it
              reaches (the later one on equal IoU; a crowd box may be taken repeatedly; an ignored box only
              when no regular one qualifies); unmatched detections outside the area range are ignored.
  accumulate  per (category, area range, detection cap): all detections by score (stable), cumulative TP / FP
              over the non-ignored ones, precision made monotone from the right and sampled at 101 recall
              levels.
  summarize   the twelve numbers of `COCOeval.stats`, printed in pycocotools' format.

One quirk is kept for result identity: matches are recorded as ground-truth ids, so an annotation with id 0
counts as "unmatched" exactly as it does there.

iouType 'segm' is what the reference evaluates rotated boxes with (infer.py:166: the four corners as a polygon,
rasterised to run-length masks by the package's C mask API, IoU counted in pixels).  Here the IoU of two such polygons
is computed EXACTLY (convex clipping, float64) instead of on rasterised masks: the two differ by boundary pixels --
~perimeter / area relative, a fraction of a percent for boxes of thousands of pixels -- so the rotated AP is this
module's own metric, not a bit-for-bit restatement.  Supported: one convex polygon per annotation (what
RotatedCocoDataset annotations and rotated detections are); RLE masks, multi-part and non-convex polygons raise.

Matching is vectorised over the ten IoU thresholds; the loop that remains is over the detections of one image.
"""
import numpy as np


class Params:
    def __init__(self):
        self.iouThrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.recThrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ['all', 'small', 'medium', 'large']
        self.imgIds, self.catIds = [], []


def box_iou(dt, gt, crowd):
    """[D, 4] x [G, 4] boxes as x, y, w, h (float64) -> [D, G]; against crowd boxes the union is the detection.
    (A fifth field -- the angle of a rotated box -- is ignored: the axis-aligned extent before the turn.)"""
    dt = np.asarray([list(b)[:4] for b in dt], np.float64).reshape(-1, 4)
    gt = np.asarray([list(b)[:4] for b in gt], np.float64).reshape(-1, 4)
    w = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    h = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.where((w > 0) & (h > 0), w * h, 0.0)
    d_area, g_area = (dt[:, 2] * dt[:, 3])[:, None], (gt[:, 2] * gt[:, 3])[None, :]
    union = np.where(np.asarray(crowd, bool)[None, :], d_area, d_area + g_area - inter)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(inter > 0, inter / union, 0.0)


def _polygon(ann):
    """The single convex polygon of an annotation as a counter-clockwise [K, 2] float64 array."""
    seg = ann.get('segmentation')
    if not seg and len(ann.get('bbox', ())) in (4, 5):                           # a box without its polygon: derive it the way
        from .infer import rotated_corners                                       # the detections' polygons are made; a plain
        x, y, w, h, theta = (np.asarray([v], np.float64) for v in (list(ann['bbox']) + [0.0])[:5])   # box has theta = 0 (data.py)
        seg = rotated_corners(x, y, w, h, theta).tolist()
    if not isinstance(seg, (list, tuple)) or len(seg) != 1 or len(seg[0]) < 6 or len(seg[0]) % 2:
        raise NotImplementedError("iouType 'segm': annotation {} is not ONE polygon (RLE masks and multi-part polygons are "
                                  "not provided, see the module docstring)".format(ann.get('id')))
    pts = np.asarray(seg[0], np.float64).reshape(-1, 2)
    nxt = np.roll(pts, -1, axis=0)
    signed = 0.5 * float(np.sum(pts[:, 0] * nxt[:, 1] - nxt[:, 0] * pts[:, 1]))
    if signed < 0:
        pts, nxt = pts[::-1].copy(), None
    edge = np.roll(pts, -1, axis=0) - pts
    turn = edge[:, 0] * np.roll(edge, -1, axis=0)[:, 1] - edge[:, 1] * np.roll(edge, -1, axis=0)[:, 0]
    if np.any(turn < -1e-9 * max(abs(signed), 1.0)):
        raise NotImplementedError("iouType 'segm': annotation {} is not convex".format(ann.get('id')))
    return pts, abs(signed)


def _clip_convex(subject, clipper):
    """Sutherland-Hodgman: the part of polygon `subject` inside the convex, counter-clockwise `clipper` ([K, 2] arrays)."""
    out = [tuple(p) for p in subject]
    for i in range(len(clipper)):
        if not out:
            break
        ax, ay = clipper[i]
        bx, by = clipper[(i + 1) % len(clipper)]
        ex, ey = bx - ax, by - ay
        src, out = out, []
        side = [ex * (py - ay) - ey * (px - ax) for px, py in src]           # > 0: left of the edge = inside
        for j, (px, py) in enumerate(src):
            k = (j + 1) % len(src)
            sp, sq = side[j], side[k]
            if sp >= 0:
                out.append((px, py))
            if (sp > 0 > sq) or (sp < 0 < sq):
                t = sp / (sp - sq)
                out.append((px + (src[k][0] - px) * t, py + (src[k][1] - py) * t))
    return out


def polygon_iou(dt, gt, crowd):
    """[D] x [G] annotations with one convex polygon each -> [D, G] IoU (float64); against crowd regions the union is the
    detection (maskUtils.iou's iscrowd rule)."""
    d_poly, g_poly = [_polygon(a) for a in dt], [_polygon(a) for a in gt]
    iou = np.zeros((len(dt), len(gt)))
    for i, (dp, da) in enumerate(d_poly):
        d_lo, d_hi = dp.min(0), dp.max(0)
        for j, (gp, ga) in enumerate(g_poly):
            if np.any(d_hi < gp.min(0)) or np.any(gp.max(0) < d_lo):         # disjoint extents
                continue
            piece = _clip_convex(dp, gp)
            if len(piece) < 3:
                continue
            q = np.asarray(piece)
            nxt = np.roll(q, -1, axis=0)
            inter = 0.5 * abs(float(np.sum(q[:, 0] * nxt[:, 1] - nxt[:, 0] * q[:, 1])))
            union = da if crowd[j] else da + ga - inter
            if inter > 0 and union > 0:
                iou[i, j] = inter / union
    return iou


class COCOeval:
    def __init__(self, cocoGt, cocoDt, iouType='bbox'):
        if iouType not in ('bbox', 'segm'):
            raise NotImplementedError("iouType '{}': 'bbox' and 'segm' are provided (see the module docstring)".format(iouType))
        self.iouType = iouType
        self.cocoGt, self.cocoDt = cocoGt, cocoDt
        self.params = Params()
        self.params.imgIds = sorted(cocoGt.getImgIds())
        self.params.catIds = sorted(cocoGt.getCatIds())
        self.evalImgs, self.eval, self.stats = [], {}, []

    # -- evaluate -----------------------------------------------------------------------------------------
    def _group(self, index, img_ids, cat_ids):
        groups = {}
        for ann in index.loadAnns(index.getAnnIds(imgIds=list(img_ids))):
            if ann['category_id'] in cat_ids:
                groups.setdefault((ann['image_id'], ann['category_id']), []).append(ann)
        return groups

    def evaluate(self):
        p = self.params
        p.imgIds, p.catIds = list(np.unique(p.imgIds)), list(np.unique(p.catIds))
        p.maxDets = sorted(p.maxDets)
        cats = set(p.catIds)
        gts = self._group(self.cocoGt, p.imgIds, cats)
        dts = self._group(self.cocoDt, p.imgIds, cats)
        cap = p.maxDets[-1]
        self.evalImgs = []                                                   # order: category, area range, image
        per_pair = {}
        for key in set(gts) | set(dts):
            gt, dt = gts.get(key, []), dts.get(key, [])
            order = np.argsort([-d['score'] for d in dt], kind='mergesort')[:cap]
            dt = [dt[i] for i in order]
            crowd = np.array([bool(g.get('iscrowd', 0)) for g in gt], bool)
            if not (gt and dt):
                iou = np.zeros((len(dt), len(gt)))
            elif self.iouType == 'segm':
                iou = polygon_iou(dt, gt, crowd)
            else:
                iou = box_iou([d['bbox'] for d in dt], [g['bbox'] for g in gt], crowd)
            per_pair[key] = (gt, dt, crowd, iou)
        for cat in p.catIds:
            for rng in p.areaRng:
                for img in p.imgIds:
                    pair = per_pair.get((img, cat))
                    self.evalImgs.append(None if pair is None else self._match(pair, rng, cap))
        return self.evalImgs

    def _match(self, pair, rng, cap):
        gt, dt, crowd, iou = pair
        thrs = self.params.iouThrs
        T, D, G = len(thrs), len(dt), len(gt)
        ignore = np.array([bool(g.get('iscrowd', 0)) or g['area'] < rng[0] or g['area'] > rng[1] for g in gt], bool)
        g_order = np.argsort(ignore, kind='mergesort')                       # regular boxes first, file order within
        ignore, crowd_s = ignore[g_order], crowd[g_order]
        iou = iou[:, g_order] if G and D else iou
        gt_ids = np.array([gt[i]['id'] for i in g_order], dtype=np.int64)
        dt_ids = np.array([d['id'] for d in dt], dtype=np.int64)
        gtm = np.zeros((T, G), dtype=np.int64)
        dtm = np.zeros((T, D), dtype=np.int64)
        dt_ignore = np.zeros((T, D), dtype=bool)
        if G and D:
            floor = np.minimum(thrs, 1 - 1e-10)[:, None]                     # [T, 1]
            n_regular = int((~ignore).sum())
            for d in range(D):
                free = ~((gtm > 0) & ~crowd_s[None, :])                      # [T, G]
                reach = np.where(free & (iou[d][None, :] >= floor), iou[d][None, :], -1.0)
                choice = np.full(T, -1)
                for lo, hi in ((0, n_regular), (n_regular, G)):              # regular boxes, then ignored ones
                    if hi <= lo:
                        continue
                    part = reach[:, lo:hi]
                    best = part.max(axis=1)
                    last = hi - 1 - np.argmax(part[:, ::-1], axis=1)         # the later box on equal IoU
                    take = (choice < 0) & (best >= 0)
                    choice = np.where(take, last, choice)
                hit = np.nonzero(choice >= 0)[0]
                g = choice[hit]
                dt_ignore[hit, d] = ignore[g]
                dtm[hit, d] = gt_ids[g]
                gtm[hit, g] = dt_ids[d]
        area = np.array([d['area'] for d in dt], dtype=np.float64)
        outside = (area < rng[0]) | (area > rng[1])
        dt_ignore |= (dtm == 0) & outside[None, :]
        return {'dtMatches': dtm, 'dtIgnore': dt_ignore, 'gtIgnore': ignore,
                'dtScores': np.array([d['score'] for d in dt], dtype=np.float64), 'maxDet': cap}

    # -- accumulate ---------------------------------------------------------------------------------------
    def accumulate(self):
        if not self.evalImgs:
            raise RuntimeError('Please run evaluate() first')
        p = self.params
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds), len(p.areaRng), len(p.maxDets)
        I = len(p.imgIds)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        scores = -np.ones((T, R, K, A, M))
        for k in range(K):
            for a in range(A):
                cell = [e for e in self.evalImgs[(k * A + a) * I:(k * A + a + 1) * I] if e is not None]
                if not cell:
                    continue
                gt_ignore = np.concatenate([e['gtIgnore'] for e in cell])
                n_regular = int(np.count_nonzero(~gt_ignore))
                if n_regular == 0:
                    continue
                for m, cap in enumerate(p.maxDets):
                    s = np.concatenate([e['dtScores'][:cap] for e in cell])
                    order = np.argsort(-s, kind='mergesort')
                    s = s[order]
                    matched = np.concatenate([e['dtMatches'][:, :cap] for e in cell], axis=1)[:, order] != 0
                    ignored = np.concatenate([e['dtIgnore'][:, :cap] for e in cell], axis=1)[:, order]
                    tp = np.cumsum(matched & ~ignored, axis=1).astype(np.float64)
                    fp = np.cumsum(~matched & ~ignored, axis=1).astype(np.float64)
                    nd = tp.shape[1]
                    for t in range(T):
                        rc = tp[t] / n_regular
                        pr = tp[t] / (fp[t] + tp[t] + np.spacing(1))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        q, ss = np.zeros(R), np.zeros(R)
                        if nd:
                            pr = np.maximum.accumulate(pr[::-1])[::-1]       # precision envelope
                            at = np.searchsorted(rc, p.recThrs, side='left')
                            ok = at < nd
                            q[ok], ss[ok] = pr[at[ok]], s[at[ok]]
                        precision[t, :, k, a, m] = q
                        scores[t, :, k, a, m] = ss
        self.eval = {'params': p, 'counts': [T, R, K, A, M], 'precision': precision, 'recall': recall, 'scores': scores}
        return self.eval

    # -- summarize ----------------------------------------------------------------------------------------
    def _summary(self, ap, iou_thr=None, area='all', max_dets=100, out=print):
        p = self.params
        a, m = p.areaRngLbl.index(area), p.maxDets.index(max_dets)
        table = self.eval['precision'] if ap else self.eval['recall']
        if iou_thr is not None:
            table = table[np.where(iou_thr == p.iouThrs)[0]]
        table = table[..., a, m]
        valid = table[table > -1]
        value = float(np.mean(valid)) if valid.size else -1.0
        span = '{:0.2f}:{:0.2f}'.format(p.iouThrs[0], p.iouThrs[-1]) if iou_thr is None else '{:0.2f}'.format(iou_thr)
        out(' {:<18} {} @[ IoU={:<9} | area={:>6s} | maxDets={:>3d} ] = {:0.3f}'.format(
            'Average Precision' if ap else 'Average Recall', '(AP)' if ap else '(AR)', span, area, max_dets, value))
        return value

    def summarize(self, out=print):
        if not self.eval:
            raise RuntimeError('Please run accumulate() first')
        top = self.params.maxDets[-1]
        rows = [(1, None, 'all', top), (1, .5, 'all', top), (1, .75, 'all', top),
                (1, None, 'small', top), (1, None, 'medium', top), (1, None, 'large', top),
                (0, None, 'all', self.params.maxDets[0]), (0, None, 'all', self.params.maxDets[1]), (0, None, 'all', top),
                (0, None, 'small', top), (0, None, 'medium', top), (0, None, 'large', top)]
        self.stats = np.array([self._summary(ap, thr, area, cap, out) for ap, thr, area, cap in rows])
        return self.stats
