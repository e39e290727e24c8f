"""Generate tests/golden/*.npz by running the REFERENCE's own odtk/box.py (build container only).

    python -m oracle.gen_golden          # from the repo root; needs /root/reference

TEST INFRASTRUCTURE ONLY.  Each fixture stores the exact input arrays and the outputs the
unmodified reference produced for them (via oracle/ref_loader.py: stub odtk._C + legacy
int-division shim).  Inputs are tie-free among candidates (asserted), so the reference's
implementation-defined torch.topk / torch.sort tie order cannot leak into the fixtures.
The fixtures travel to the GPU box; /root/reference does not.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'retinanet-examples_amd'))

from oracle import ref_loader  # noqa: E402
from odtk import synthetic     # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]


def _np(t):
    return t.detach().cpu().numpy()


def _assert_tie_free(scores, threshold):
    flat = scores.reshape(scores.shape[0], -1)
    for b in range(flat.shape[0]):
        v = flat[b][flat[b] >= threshold]
        assert v.unique().numel() == v.numel(), 'ties among candidates'


def decode_case(name, batch, classes, h, w, stride, kind, seed, threshold, top_n, tweak=None):
    anchors = ref_loader.ref_generate_anchors(stride, RATIOS, SCALES)
    logits, deltas = synthetic.make_level(batch, anchors.shape[0], classes, h, w, kind, seed)
    scores = logits.sigmoid()
    if tweak:
        scores, deltas = tweak(scores, deltas)
    scores = synthetic.make_unique_scores(scores, threshold)
    _assert_tie_free(scores, threshold)
    out = ref_loader.ref_decode(scores, deltas, stride, threshold, top_n, anchors)
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'),
                        kind='decode', cls=_np(scores), box=_np(deltas), anchors=_np(anchors),
                        stride=stride, threshold=np.float64(threshold), top_n=top_n,
                        out_scores=_np(out[0]), out_boxes=_np(out[1]), out_classes=_np(out[2]))
    k = int((out[0] > 0).sum())
    print('%-28s cls %s  candidates kept %d' % (name, tuple(scores.shape), k))


def pipeline_case(name, batch, classes, height, width, kind, seed, threshold=0.05, top_n=1000,
                  nms=0.5, detections=100):
    """model.py:153-165: decode x5 -> cat -> nms, all by the reference."""
    cls, box, strides = synthetic.pyramid(batch, 9, classes, height, width, kind, seed, threshold=threshold)
    anchors = {s: ref_loader.ref_generate_anchors(s, RATIOS, SCALES) for s in strides}
    decoded = [ref_loader.ref_decode(c, b, s, threshold, top_n, anchors[s]) for c, b, s in zip(cls, box, strides)]
    cat = [torch.cat(t, 1) for t in zip(*decoded)]
    _assert_tie_free(cat[0], 1e-30)
    out = ref_loader.ref_nms(*cat, nms, detections)
    payload = dict(kind='pipeline', strides=np.array(strides), threshold=np.float64(threshold), top_n=top_n,
                   nms=np.float64(nms), detections=detections,
                   cat_scores=_np(cat[0]), cat_boxes=_np(cat[1]), cat_classes=_np(cat[2]),
                   out_scores=_np(out[0]), out_boxes=_np(out[1]), out_classes=_np(out[2]))
    for i, s in enumerate(strides):
        payload['cls%d' % i] = _np(cls[i])
        payload['box%d' % i] = _np(box[i])
        payload['anchors%d' % i] = _np(anchors[s])
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **payload)
    print('%-28s nms in %d -> out %d' % (name, int((cat[0] > 0).sum()), int((out[0] > 0).sum())))


def nms_case(name, batch, count, seed, nms, detections, n_classes=3, zero_frac=0.3):
    """Direct NMS input: clustered random boxes, unique positive scores, some zero padding."""
    g = torch.Generator().manual_seed(seed)
    centres = torch.rand(batch, 12, 2, generator=g) * 400 + 50
    which = torch.randint(0, 12, (batch, count), generator=g)
    ctr = torch.gather(centres, 1, which[..., None].expand(-1, -1, 2)) + torch.randn(batch, count, 2, generator=g) * 12
    wh = torch.rand(batch, count, 2, generator=g) * 80 + 20
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 2).clamp(0, 511)
    scores = torch.rand(batch, count, generator=g) * 0.95 + 0.05
    scores[torch.rand(batch, count, generator=g) < zero_frac] = 0
    scores = synthetic.make_unique_scores(scores, 1e-30)
    classes = torch.randint(0, n_classes, (batch, count), generator=g).float()
    out = ref_loader.ref_nms(scores, boxes, classes, nms, detections)
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), kind='nms',
                        scores=_np(scores), boxes=_np(boxes), classes=_np(classes),
                        nms=np.float64(nms), detections=detections,
                        out_scores=_np(out[0]), out_boxes=_np(out[1]), out_classes=_np(out[2]))
    print('%-28s in %d -> out %d' % (name, int((scores > 0).sum()), int((out[0] > 0).sum())))


def anchors_case():
    payload = {}
    for s in (8, 16, 32, 64, 128):
        payload['s%d' % s] = _np(ref_loader.ref_generate_anchors(s, RATIOS, SCALES))
    ang = [-np.pi / 6, 0, np.pi / 6]
    for s in (8, 16, 32, 64, 128):
        ax, rot = ref_loader.ref_generate_anchors_rotated(s, RATIOS, SCALES, ang)
        payload['rot_axis_s%d' % s] = _np(ax)
        payload['rot_pts_s%d' % s] = _np(rot)
    np.savez_compressed(os.path.join(GOLDEN, 'anchors.npz'), **payload)
    print('anchors.npz')


def snap_case(name, seed, stride, size, n_boxes, classes=10, ious=(0.4, 0.5)):
    """reference box.py:134-189 on random ground-truth boxes (x, y, w, h, class)."""
    import warnings
    warnings.filterwarnings('ignore')
    g = torch.Generator().manual_seed(seed)
    anchors = ref_loader.ref_generate_anchors(stride, RATIOS, SCALES)
    xy = torch.rand(n_boxes, 2, generator=g) * torch.tensor([size[0] * 0.7, size[1] * 0.7])
    wh = torch.rand(n_boxes, 2, generator=g) * torch.tensor([size[0] * 0.5, size[1] * 0.5]) + 8
    cls = torch.randint(0, classes, (n_boxes, 1), generator=g).float()
    boxes = torch.cat([xy.floor(), wh.floor(), cls], 1)
    out = ref_loader.reference_box().snap_to_anchors(boxes, list(size), stride, anchors, classes, 'cpu', list(ious))
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), kind='snap', boxes=_np(boxes), size=np.array(size),
                        stride=stride, anchors=_np(anchors), classes=classes, ious=np.array(ious),
                        cls_target=_np(out[0]), box_target=_np(out[1]), depth=_np(out[2]))
    print('%-28s fg %d ignore %d' % (name, int((out[2] > 0).sum()), int((out[2] < 0).sum())))


def main():
    assert ref_loader.available(), 'needs /root/reference'
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(1)
    anchors_case()

    # K < top_n, K > top_n, different strides / aspect ratios / class counts
    decode_case('decode_sparse_p5', 2, 80, 13, 20, 32, 'sparse', 11, 0.05, 1000)
    decode_case('decode_dense_p5_topn100', 2, 80, 13, 20, 32, 'dense', 12, 0.05, 100)
    decode_case('decode_dense_p4', 1, 20, 25, 40, 16, 'dense', 13, 0.05, 1000)
    decode_case('decode_clustered_p6', 3, 20, 7, 10, 64, 'clustered', 14, 0.3, 50)

    def below(scores, deltas):
        return scores.clamp(max=0.04), deltas
    decode_case('decode_none_above', 2, 8, 5, 7, 128, 'sparse', 15, 0.05, 20, below)

    def at_threshold(scores, deltas):
        s = scores.clone()
        s.view(s.shape[0], -1)[:, ::97] = torch.tensor(0.05, dtype=torch.float32)   # == float32(0.05): kept by >=
        s.view(s.shape[0], -1)[:, 1::97] = torch.nextafter(torch.tensor(0.05, dtype=torch.float32), torch.tensor(0.0))
        return s, deltas
    # ties at exactly the threshold value are made unique by make_unique_scores (one stays == thr)
    decode_case('decode_at_threshold', 1, 8, 6, 9, 8, 'sparse', 16, 0.05, 1000, at_threshold)

    def off_image(scores, deltas):
        return scores, deltas * 12.0          # huge |dx|,|dy| and exp(dw) -> both clamps active
    decode_case('decode_clamped', 2, 8, 9, 11, 16, 'dense', 17, 0.05, 200, off_image)

    pipeline_case('pipeline_clustered_160x256', 2, 40, 160, 256, 'clustered', 21)
    pipeline_case('pipeline_dense_128x128', 1, 20, 128, 128, 'dense', 22, top_n=300, detections=50)

    nms_case('nms_clustered_3000', 2, 3000, 31, 0.5, 100)
    nms_case('nms_thr03_det10', 3, 500, 32, 0.3, 10)
    nms_case('nms_few', 2, 40, 33, 0.5, 100, zero_frac=0.8)

    snap_case('snap_s16_256x160', 41, 16, (256, 160), 7)
    snap_case('snap_s8_128x128', 42, 8, (128, 128), 20)
    snap_case('snap_s64_one_box', 43, 64, (256, 192), 1)
    snap_case('snap_s32_none', 44, 32, (128, 96), 0)


if __name__ == '__main__':
    main()
