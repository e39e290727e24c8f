"""BASELINE.json's full configuration (bs=8, 800x1280, 5 levels, A=9, C=80: 122.9 M scores per
batch) is far too big for the CPU oracle, so parity at that size is checked through size-independent
properties and through torch's own GPU ops as an independent implementation:
  * per (image, level): emitted scores are exactly torch.topk's values (bit for bit), sorted,
    and their count is min(top_n, #{score >= thr});
  * every emitted index points at its score, is unique, and yields the emitted class;
  * boxes lie inside the level's clamp window;
  * detect == nms(decode_levels); NMS is idempotent; kept boxes of one class never overlap > thr.
"""
import pytest
import torch

from odtk import _C, box, synthetic

pytestmark = pytest.mark.gpu

RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]
STRIDES = [8, 16, 32, 64, 128]


def full_heads(kind, dtype, logits, channels_last, batch=8, seed=4321):
    g = torch.Generator(device='cuda').manual_seed(seed)
    cls, dl = [], []
    for (h, w) in synthetic.level_shapes(800, 1280, STRIDES):
        c = torch.randn((batch, 720, h, w), generator=g, device='cuda') * synthetic.SIGMA[kind] + synthetic.LOGIT_PRIOR
        if not logits:
            c = c.sigmoid()
        d = torch.randn((batch, 36, h, w), generator=g, device='cuda') * 0.2
        c, d = c.to(dtype), d.to(dtype)
        if channels_last:
            c, d = c.contiguous(memory_format=torch.channels_last), d.contiguous(memory_format=torch.channels_last)
        cls.append(c)
        dl.append(d)
    return cls, dl


def pairwise_iou_plus1(b):
    x1 = torch.max(b[:, None, 0], b[None, :, 0]); y1 = torch.max(b[:, None, 1], b[None, :, 1])
    x2 = torch.min(b[:, None, 2], b[None, :, 2]); y2 = torch.min(b[:, None, 3], b[None, :, 3])
    inter = (x2 - x1 + 1).clamp(0) * (y2 - y1 + 1).clamp(0)
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    return inter / (area[:, None] + area[None, :] - inter)


@pytest.mark.parametrize('kind,dtype,logits,channels_last', [
    ('sparse', torch.float32, False, False),        # the reference boundary: fp32 NCHW scores
    ('sparse', torch.bfloat16, True, True),         # the fused path Model.forward uses
    ('dense', torch.bfloat16, True, True),          # 5 % candidates: radix descent on 570 k keys
], ids=['fp32-nchw-scores', 'bf16-nhwc-logits', 'bf16-nhwc-logits-dense'])
def test_full_size_properties(kind, dtype, logits, channels_last):
    top_n, thr, B = 1000, 0.05, 8
    cls, dl = full_heads(kind, dtype, logits, channels_last)
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in STRIDES}
    out = _C.decode_levels(cls, dl, [anchors[s] for s in STRIDES], STRIDES, thr, top_n, False,
                           return_indices=True, logits=logits)
    scores, boxes, classes, indices = out
    assert scores.shape == (B, 5 * top_n) and boxes.shape == (B, 5 * top_n, 4)
    for l, c in enumerate(cls):
        h, w = c.shape[2:]
        # what the op sees, materialised by torch in canonical NCHW order
        s_ref = (c.sigmoid() if logits else c).contiguous().float().view(B, -1)
        sl = slice(l * top_n, (l + 1) * top_n)
        lv_s, lv_i, lv_c, lv_b = scores[:, sl], indices[:, sl].long(), classes[:, sl], boxes[:, sl]
        n_cand = (s_ref >= thr).sum(1)
        n_out = (lv_i >= 0).sum(1)
        assert torch.equal(n_out, n_cand.clamp(max=top_n)), 'level %d: survivor count' % l
        k = int(n_out.min())
        if k:
            ref_top = torch.topk(s_ref, k, dim=1).values
            assert torch.equal(lv_s[:, :k], ref_top), 'level %d: top-k values' % l
        for b in range(B):
            nb = int(n_out[b])
            sb, ib = lv_s[b, :nb], lv_i[b, :nb]
            assert torch.all(sb[:-1] >= sb[1:]) and torch.all(sb >= thr)
            assert torch.all(lv_s[b, nb:] == 0) and torch.all(lv_i[b, nb:] == -1) and torch.all(lv_b[b, nb:] == 0)
            assert ib.unique().numel() == nb
            assert torch.equal(s_ref[b, ib], sb)
            assert torch.equal(((ib // (h * w)) % 80).float(), lv_c[b, :nb])
            ties = sb[:-1] == sb[1:]
            assert torch.all(ib[:-1][ties] < ib[1:][ties]), 'ties must be in ascending index order'
        lim = torch.tensor([w * STRIDES[l] - 1, h * STRIDES[l] - 1] * 2, device='cuda', dtype=torch.float32)
        assert torch.all(lv_b >= 0) and torch.all(lv_b <= lim)

    det = box.detect(cls, dl, STRIDES, anchors, thr, top_n, 0.5, 100, logits=logits)
    via = _C.nms(scores, boxes, classes, 0.5, 100)
    for a, b_ in zip(det, via):
        assert torch.equal(a, b_)                                   # composition
    again = _C.nms(det[0], det[1], det[2], 0.5, 100)
    for a, b_ in zip(det, again):
        assert torch.equal(a, b_)                                   # idempotence
    for b in range(B):
        n = int((det[0][b] > 0).sum())
        assert n == 100
        iou = pairwise_iou_plus1(det[1][b, :n])
        same = det[2][b, :n, None] == det[2][b, None, :n]
        off = ~torch.eye(n, dtype=torch.bool, device='cuda')
        assert not torch.any((iou > 0.5) & same & off)
        assert torch.all(det[0][b, :n - 1] >= det[0][b, 1:n])


def test_full_size_saturated_and_empty():
    """All-equal scores at full size: the lowest 1000 indices of every level must come out, in order
    (ties by ascending index), through the sub-list overflow -> raw-score path; and nothing at all."""
    B = 2
    cls = [torch.ones((B, 720, h, w), device='cuda') for (h, w) in synthetic.level_shapes(800, 1280, STRIDES)]
    dl = [torch.zeros((B, 36, h, w), device='cuda') for (h, w) in synthetic.level_shapes(800, 1280, STRIDES)]
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in STRIDES}
    out = _C.decode_levels(cls, dl, [anchors[s] for s in STRIDES], STRIDES, 0.05, 1000, False, return_indices=True)
    expect = torch.arange(1000, device='cuda', dtype=torch.int32).repeat(B, 5)
    assert torch.equal(out[3], expect)
    assert torch.all(out[0] == 1.0)
    for c in cls:
        c.fill_(0.01)
    out = _C.decode_levels(cls, dl, [anchors[s] for s in STRIDES], STRIDES, 0.05, 1000, False, return_indices=True)
    assert torch.all(out[3] == -1) and torch.all(out[0] == 0) and torch.all(out[1] == 0)
    det = box.detect(cls, dl, STRIDES, anchors)
    assert torch.all(det[0] == 0)
