#!/usr/bin/env python
"""Golden vectors for the rotated path, produced by the reference's OWN device code compiled for the CPU
(oracle/ref_build/build_ref.py -> oracle/_ref/libodtk_ref_native.so): tests/golden/rotated_ref_*.npz.
Run in the build container (needs /root/reference to build the library); the fixtures travel instead of it.

    python oracle/gen_golden_native.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_native                                   # noqa: E402
from oracle.ref_build import build_ref                           # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def quads(r, n, lo, hi, size):
    ctr = r.uniform(lo, hi, (n, 2))
    wh = r.uniform(size[0], size[1], (n, 2))
    th = r.uniform(-1.5, 1.5, n)
    c, s = np.cos(th), np.sin(th)
    dx = np.stack([-wh[:, 0], wh[:, 0], wh[:, 0], -wh[:, 0]], 1) / 2
    dy = np.stack([-wh[:, 1], -wh[:, 1], wh[:, 1], wh[:, 1]], 1) / 2
    x = dx * c[:, None] - dy * s[:, None] + ctr[:, 0:1]
    y = dy * c[:, None] + dx * s[:, None] + ctr[:, 1:2]
    return np.stack([x, y], 2).reshape(n, 8).astype(np.float32)


def boxes6(r, k, span, size, unit=True):
    ctr = r.uniform(20, span, (k, 2))
    wh = r.uniform(size[0], size[1], (k, 2))
    th = r.uniform(-1.5, 1.5, k)
    scale = 1.0 if unit else r.uniform(0.6, 1.4, k)               # the network does not normalise (sin, cos)
    return np.concatenate([ctr - wh / 2, ctr + wh / 2, (np.sin(th) * scale)[:, None], (np.cos(th) * scale)[:, None]],
                          1).astype(np.float32)


def main():
    build_ref.build()
    r = np.random.default_rng(20240924)
    # pairwise IoU: ground-truth quads x anchor quads (training-side target assignment)
    for name, n, m, lo, hi, size in (('a', 23, 300, 40, 400, (8, 160)), ('b', 1, 64, 100, 200, (30, 90)),
                                     ('c', 40, 40, 0, 120, (4, 60))):
        b, a = quads(r, n, lo, hi, size), quads(r, m, lo, hi, size)
        if name == 'c':
            a[:10] = b[:10]                                        # identical quads: the 0.001 pad rule
            a[10:14, :2] = b[10:14, :2]                            # one shared corner
        np.savez_compressed(os.path.join(GOLDEN, 'rotated_ref_iou_%s.npz' % name), boxes=b, anchors=a,
                            iou=ref_native.iou_pairs(b, a))
    # rotated NMS, one image each
    for name, k, span, size, n_cls, thr, ndet, unit in (('a', 600, 320, (4, 90), 3, 0.5, 100, True),
                                                        ('b', 1000, 500, (8, 200), 80, 0.3, 100, True),
                                                        ('c', 300, 150, (10, 120), 1, 0.0, 300, True),
                                                        ('d', 800, 400, (6, 150), 2, 0.7, 50, False)):
        bx = boxes6(r, k, span, size, unit)
        sc = (r.permutation(k).astype(np.float32) + 1) / k        # tie-free
        sc[r.random(k) < 0.15] = 0
        cl = r.integers(0, n_cls, k).astype(np.float32)
        s, b, c, idx = ref_native.nms_rotate(sc, bx, cl, thr, ndet)
        np.savez_compressed(os.path.join(GOLDEN, 'rotated_ref_nms_%s.npz' % name), scores=sc, boxes=bx, classes=cl,
                            thresh=np.float32(thr), ndet=np.int32(ndet), out_scores=s, out_boxes=b, out_classes=c,
                            out_index=idx)
    # axis-aligned NMS through the reference's CUDA nms_kernel (csrc/cuda/nms.cu:44-80): the CPU path
    # (odtk/box.py) is normative for this repo, this pins that both of the reference's paths agree with it
    for name, k, span, size, n_cls, thr, ndet in (('a', 1500, 600, (8, 200), 80, 0.5, 100), ('b', 400, 200, (10, 150), 2, 0.3, 400)):
        ctr = r.uniform(20, span, (k, 2))
        wh = r.uniform(size[0], size[1], (k, 2))
        bx = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
        sc = (r.permutation(k).astype(np.float32) + 1) / k
        sc[r.random(k) < 0.15] = 0
        cl = r.integers(0, n_cls, k).astype(np.float32)
        s, b, c, idx = ref_native.nms_axis(sc, bx, cl, thr, ndet)
        np.savez_compressed(os.path.join(GOLDEN, 'axis_ref_nms_%s.npz' % name), scores=sc, boxes=bx, classes=cl,
                            thresh=np.float32(thr), ndet=np.int32(ndet), out_scores=s, out_boxes=b, out_classes=c,
                            out_index=idx)
    # decode: the reference's CUDA gather + box lambdas (decode.cu:121-159, decode_rotate.cu:116-167) applied to the
    # indices the CPU-convention selection keeps.  Index decomposition, gather layout and sin/cos passthrough must be
    # exact; the box differs from the CPU path only by documented conventions (one- vs two-sided clamp, the order of
    # the centre sum, float exp) -- tests apply the two-sided clamp and a 1e-4 tolerance.
    import math
    sys.path.insert(0, os.path.join(ROOT, 'retinanet-examples_amd'))
    from odtk import box as box_ops
    from oracle import c_oracle
    ratios, scales = [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)]
    for name, rotated, n_cls, h, w, stride, thr, top_n in (('axis', False, 7, 19, 23, 16, 0.4, 500), ('rotated', True, 5, 11, 14, 32, 0.5, 300)):
        if rotated:
            anchors = box_ops.generate_anchors_rotated(stride, ratios, scales, [-math.pi / 6, 0, math.pi / 6])[0].numpy()
        else:
            anchors = box_ops.generate_anchors(stride, ratios, scales).numpy()
        a, nb = anchors.shape[0], 6 if rotated else 4
        cls = r.random((a * n_cls, h, w)).astype(np.float32)
        dl = (r.standard_normal((a * nb, h, w)) * 0.3).astype(np.float32)
        idx = c_oracle.decode(cls[None], dl[None], stride, thr, top_n, anchors, rotated=rotated)[3][0]
        idx = idx[idx >= 0].astype(np.int32)
        s, b, c = ref_native.decode_gather(idx, cls, dl, stride, anchors, n_cls, rotated)
        np.savez_compressed(os.path.join(GOLDEN, 'decode_ref_%s.npz' % name), cls=cls, deltas=dl, anchors=anchors,
                            stride=np.int32(stride), thresh=np.float32(thr), top_n=np.int32(top_n), num_classes=np.int32(n_cls),
                            indices=idx, out_scores=s, out_boxes=b, out_classes=c)
    # rotated target assignment: the reference's own snap_to_anchors_rotated (odtk/box.py:192-252) run on the CPU
    # with its `iou_cuda` bound to its own iou kernel (oracle/ref_loader.py:ref_snap_to_anchors_rotated)
    import warnings
    import torch
    from oracle import ref_loader
    warnings.filterwarnings('ignore')
    angles = [-math.pi / 6, 0, math.pi / 6]
    for name, seed, stride, size, n, n_cls in (('a', 61, 16, (160, 128), 6, 7), ('b', 62, 32, (256, 192), 12, 80), ('c', 63, 8, (64, 48), 0, 5)):
        g = torch.Generator().manual_seed(seed)
        anchors = ref_loader.ref_generate_anchors_rotated(stride, ratios, scales, angles)
        xy = torch.rand(n, 2, generator=g) * torch.tensor([size[0] * 0.7, size[1] * 0.7])
        wh = torch.rand(n, 2, generator=g) * torch.tensor([size[0] * 0.4, size[1] * 0.4]) + 12
        th = (torch.rand(n, 1, generator=g) - 0.5) * 1.4
        boxes = torch.cat([xy, wh, th, torch.randint(0, n_cls, (n, 1), generator=g).float()], 1)
        cls_t, box_t, depth = ref_loader.ref_snap_to_anchors_rotated(boxes, list(size), stride, anchors, n_cls, [0.4, 0.5])
        np.savez_compressed(os.path.join(GOLDEN, 'snaprot_ref_%s.npz' % name), boxes=boxes.numpy(), size=np.array(size),
                            stride=np.int32(stride), classes=np.int32(n_cls), ious=np.array([0.4, 0.5], np.float32),
                            cls_target=cls_t.numpy(), box_target=box_t.numpy(), depth=depth.numpy())
    print('wrote', sorted(f for f in os.listdir(GOLDEN) if '_ref_' in f))


if __name__ == '__main__':
    main()
