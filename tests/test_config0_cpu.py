"""BASELINE config 0: "ResNet18FPN, 1 synthetic 3x512x512 image, pure-PyTorch CPU infer via odtk/box.py
decode+NMS (plumbing, no GPU)".  The product's CPU branch (odtk/box.py:_decode_cpu / _nms_cpu, written in
the product, not imported from oracle/) must
  * reproduce the golden fixtures = outputs of the REFERENCE's own odtk/box.py, bit for bit (boxes too: same
    torch ops on the same machine class), and
  * carry `Model.forward` on CPU tensors end to end, equal to the pinned oracle on the same head tensors.
No GPU, no libodtk_hip.so needed."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import box_oracle
from odtk import box
from odtk.model import Model

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _cases(kind):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, '*.npz'))):
        with np.load(p) as z:
            if 'kind' in z.files and str(z['kind']) == kind:
                out.append(p)
    return out


def _load(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def _bits(t, ref, what):
    a = np.ascontiguousarray(t.numpy(), dtype=np.float32).view(np.uint32)
    b = np.ascontiguousarray(ref, dtype=np.float32).view(np.uint32)
    assert a.shape == b.shape and np.array_equal(a, b), what


@pytest.mark.parametrize('path', _cases('decode'), ids=os.path.basename)
def test_cpu_decode_vs_reference_fixture(path):
    g = _load(path)
    out = box.decode(torch.from_numpy(g['cls']), torch.from_numpy(g['box']), int(g['stride']), float(g['threshold']),
                     int(g['top_n']), torch.from_numpy(g['anchors']))
    _bits(out[0], g['out_scores'], 'scores')
    _bits(out[2], g['out_classes'], 'classes')
    assert np.abs(out[1].numpy().astype(np.float64) - g['out_boxes']).max() <= 1e-4   # exp(): SLEEF build vs build


@pytest.mark.parametrize('path', _cases('nms'), ids=os.path.basename)
def test_cpu_nms_vs_reference_fixture(path):
    g = _load(path)
    out = box.nms(torch.from_numpy(g['scores']), torch.from_numpy(g['boxes']), torch.from_numpy(g['classes']),
                  float(g['nms']), int(g['detections']))
    for o, k in zip(out, ('out_scores', 'out_boxes', 'out_classes')):
        _bits(o, g[k], k)


def test_cpu_edge_cases():
    anchors = box.generate_anchors(8, [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)])
    cls = torch.full((2, 9 * 3, 4, 5), 0.01)
    dl = torch.zeros((2, 36, 4, 5))
    out = box.decode(cls, dl, 8, 0.05, 10, anchors)                       # nothing above the threshold
    assert all(float(o.abs().sum()) == 0 for o in out) and out[1].shape == (2, 10, 4)
    cls[1] = 0.5                                                          # every score equal: lowest indices win
    out = box.decode(cls, dl, 8, 0.05, 10, anchors)
    ref = box_oracle.decode(cls, dl, 8, 0.05, 10, anchors, return_indices=True)
    assert torch.equal(ref[3][1], torch.arange(10)) and all(torch.equal(a, b) for a, b in zip(out, ref[:3]))
    # nms: all-zero scores, ties (equal scores, identical boxes, same class -> the first position survives)
    s = torch.zeros(1, 6); b = torch.zeros(1, 6, 4); c = torch.zeros(1, 6)
    assert float(box.nms(s, b, c, 0.5, 3)[0].sum()) == 0
    s[0, :4] = torch.tensor([0.9, 0.9, 0.8, 0.8]); b[0, :4] = torch.tensor([[0., 0, 10, 10]] * 4); c[0, 2:4] = 1.0
    out, ref = box.nms(s, b, c, 0.5, 3), box_oracle.nms(s, b, c, 0.5, 3)
    assert all(torch.equal(a, r) for a, r in zip(out, ref)) and torch.equal(out[0][0], torch.tensor([0.9, 0.8, 0.0]))
    with pytest.raises(RuntimeError):
        box.nms_rotated(torch.rand(1, 4), torch.rand(1, 4, 6), torch.zeros(1, 4))


def test_config0_resnet18fpn_512_on_cpu():
    torch.manual_seed(0)
    model = Model('ResNet18FPN', classes=80)
    model.initialize(None)
    model.eval()
    x = torch.randn(1, 3, 512, 512)
    with torch.no_grad():
        cls_heads, _ = model.heads(x)
        bias = model.cls_head[-1].bias.view(1, -1, 1, 1)
        sigma = torch.cat([(c - bias).flatten() for c in cls_heads]).std()
        model.cls_head[-1].weight.mul_(0.7 / sigma)                        # the class prior alone gives zero detections
        scores, boxes, classes = model(x)                                  # eager graph + CPU branch of odtk/box.py
        cls_heads, box_heads = model.heads(x)
    assert scores.shape == (1, 100) and boxes.shape == (1, 100, 4) and classes.shape == (1, 100)
    strides = [512 // c.shape[-1] for c in cls_heads]
    assert strides == [8, 16, 32, 64, 128]
    ref = box_oracle.postprocess([c.sigmoid() for c in cls_heads], box_heads, strides, model.anchors,
                                 model.threshold, model.top_n, model.nms, model.detections)
    assert int((ref[0] > 0).sum()) > 20
    for got, want in zip((scores, boxes, classes), ref):
        assert torch.equal(got, want)


def test_infer_driver_end_to_end_on_cpu():
    """The inference driver (odtk/infer.py: Model.forward -> packed hand-off -> COCO records) with no GPU at all: config 0's
    plumbing run, two batches of two 256x256 images."""
    from odtk import infer as infer_mod
    torch.manual_seed(1)
    model = Model('ResNet18FPN', classes=12)
    model.initialize(None)
    model.eval()
    with torch.no_grad():
        model.cls_head[-1].weight.mul_(40.0)
    g = torch.Generator().manual_seed(2)
    batches = [(torch.randn(2, 3, 256, 256, generator=g), [10 + 2 * i, 11 + 2 * i], [0.5, 2.0]) for i in range(2)]
    dets = infer_mod.infer_batches(model, batches)
    assert len(dets) > 20 and {d['image_id'] for d in dets} <= {10, 11, 12, 13}
    with torch.no_grad():
        scores, boxes, classes = model(batches[0][0].contiguous(memory_format=torch.channels_last))
    first = [d for d in dets if d['image_id'] == 10]
    n = int((scores[0] > 0).sum())
    assert len(first) == n and n > 0
    assert abs(first[0]['score'] - float(scores[0, 0])) < 1e-7 and first[0]['category_id'] == int(classes[0, 0])
    x1, y1, x2, y2 = (boxes[0, 0] / 0.5).tolist()
    assert np.allclose(first[0]['bbox'], [x1, y1, x2 - x1 + 1, y2 - y1 + 1], rtol=1e-6)
    assert all(a['score'] >= b['score'] for a, b in zip(first, first[1:]))
