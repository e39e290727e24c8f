"""The fused fast path: sigmoid + dtype/layout handling inside the prefilter.

Parity argument in two steps, both checked here:
  (1) SAME BITS AS THE STRICT OP.  detect(raw logits in bf16/fp16/fp32, NCHW or channels_last,
      logits=True) must equal -- bit for bit, indices included -- the strict-parity fp32/NCHW op fed
      with what the reference pipeline materialises first: `cls_head.sigmoid()` (model.py:140) ->
      `.contiguous()` (:160) -> `.float()` (box.py:263), all done by torch on the GPU.
  (2) THE STRICT OP == THE ORACLE on those same materialised scores (as in test_gpu_parity.py).
So fused == reference-CPU semantics on the scores torch's own sigmoid kernel produces.
"""
import numpy as np
import pytest
import torch

from oracle import box_check, box_oracle
from odtk import _C, box, synthetic

pytestmark = pytest.mark.gpu

RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]


def head_logits(batch, classes, height, width, kind, seed, dtype, channels_last):
    strides = (8, 16, 32, 64, 128)
    cls, dl = [], []
    for i, (h, w) in enumerate(synthetic.level_shapes(height, width, strides)):
        lg, d = synthetic.make_level(batch, 9, classes, h, w, kind, seed + i)
        lg, d = lg.cuda().to(dtype), d.cuda().to(dtype)
        if channels_last:
            lg = lg.contiguous(memory_format=torch.channels_last)
            d = d.contiguous(memory_format=torch.channels_last)
        cls.append(lg)
        dl.append(d)
    return cls, dl, list(strides)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32], ids=['bf16', 'fp16', 'fp32'])
@pytest.mark.parametrize('channels_last', [False, True], ids=['nchw', 'nhwc'])
@pytest.mark.parametrize('kind,seed', [('sparse', 71), ('dense', 72)])
def test_fused_equals_strict_on_torch_materialised_scores(dtype, channels_last, kind, seed):
    cls, dl, strides = head_logits(2, 40, 192, 256, kind, seed, dtype, channels_last)
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}
    fused = _C.decode_levels(cls, dl, [anchors[s] for s in strides], strides, 0.05, 400, False,
                             return_indices=True, logits=True)
    # the reference pipeline's three passes, by torch
    scores = [c.sigmoid().contiguous().float() for c in cls]
    deltas = [d.contiguous().float() for d in dl]
    strict = _C.decode_levels(scores, deltas, [anchors[s] for s in strides], strides, 0.05, 400, False,
                              return_indices=True)
    for f, s, name in zip(fused, strict, ('scores', 'boxes', 'classes', 'indices')):
        assert torch.equal(f, s), name
    # (2) strict == oracle on the materialised scores
    ref = [box_oracle.decode(c.cpu(), d.cpu(), s, 0.05, 400, anchors[s], return_indices=True)
           for c, d, s in zip(scores, deltas, strides)]
    ref = [torch.cat(t, 1) for t in zip(*ref)]
    assert torch.equal(fused[3].cpu().long(), ref[3])
    assert torch.equal(fused[0].cpu(), ref[0]) and torch.equal(fused[2].cpu(), ref[2])
    box_check.check_decode(fused[1], ref[1], [c.cpu() for c in scores], [d.cpu() for d in deltas], strides, anchors, 0.05, 400,
                           ref_indices=ref[3])                    # 1e-4; beyond it only with proof (oracle/box_check.py)
    # full detect
    det = box.detect(cls, dl, strides, anchors, 0.05, 400, 0.5, 100, logits=True)
    det_ref = box_oracle.nms(ref[0], ref[1], ref[2], 0.5, 100)
    assert torch.equal(det[0].cpu(), det_ref[0]) and torch.equal(det[2].cpu(), det_ref[2])


def test_scores_without_logits_in_16bit_and_nhwc():
    """dtype / layout generality without the sigmoid: bf16 scores, channels_last."""
    cls, dl, strides = head_logits(3, 20, 128, 160, 'dense', 81, torch.float32, False)
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}
    scores16 = [c.sigmoid().bfloat16().contiguous(memory_format=torch.channels_last) for c in cls]
    dl16 = [d.bfloat16().contiguous(memory_format=torch.channels_last) for d in dl]
    out = _C.decode_levels(scores16, dl16, [anchors[s] for s in strides], strides, 0.05, 300, False, return_indices=True)
    ref = [box_oracle.decode(c.float().cpu().contiguous(), d.float().cpu().contiguous(), s, 0.05, 300, anchors[s],
                             return_indices=True) for c, d, s in zip(scores16, dl16, strides)]
    ref = [torch.cat(t, 1) for t in zip(*ref)]
    assert torch.equal(out[3].cpu().long(), ref[3])
    assert torch.equal(out[0].cpu(), ref[0]) and torch.equal(out[2].cpu(), ref[2])
    box_check.check_decode(out[1], ref[1], [c.float().cpu().contiguous() for c in scores16], [d.float().cpu().contiguous() for d in dl16],
                           strides, anchors, 0.05, 300, ref_indices=ref[3])


def test_fused_threshold_edges_and_saturation():
    """thresholds <= 0 (everything passes), >= 1 (only saturated scores), huge |logits|."""
    g = torch.Generator().manual_seed(5)
    lg = (torch.randn(2, 36, 9, 13, generator=g) * 6).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    lg[0, 0, 0, :4] = torch.tensor([80.0, -80.0, float('inf'), float('nan')]).cuda().bfloat16()
    d = (torch.randn(2, 36, 9, 13, generator=g) * 0.2).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    anchors = {16: box.generate_anchors(16, RATIOS, SCALES)}
    for thr in (0.0, -1.0, 0.5, 0.999, 1.0, 1.5):
        fused = _C.decode_levels([lg], [d], [anchors[16]], [16], thr, 64, False, return_indices=True, logits=True)
        strict = _C.decode_levels([lg.sigmoid().contiguous().float()], [d.contiguous().float()], [anchors[16]], [16],
                                  thr, 64, False, return_indices=True)
        for f, s in zip(fused, strict):
            assert torch.equal(f, s), thr


def test_model_fused_equals_reference_sequence():
    """Model.forward: the fused 3-launch path == the reference's op sequence (sigmoid, .contiguous(),
    decode x5, cat, nms) on the same weights, under bf16 autocast + channels_last (config 2 style)."""
    from odtk.model import Model
    torch.manual_seed(0)
    model = Model('ResNet18FPN', classes=20)
    model.initialize(None)
    model = model.cuda().to(memory_format=torch.channels_last).eval()
    x = torch.randn(2, 3, 256, 320, device='cuda').contiguous(memory_format=torch.channels_last)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        cls_heads, _ = model.heads(x)
        sigma = torch.cat([(c.float() - model.cls_head[-1].bias.view(1, -1, 1, 1)).flatten() for c in cls_heads]).std()
    with torch.no_grad():
        model.cls_head[-1].weight.mul_(1.0 / sigma)          # make detections exist (class prior = 0.01)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):   # fresh context: autocast caches casts
        # two forward passes of the backbone are not bit-reproducible run to run (library conv
        # kernels), so both post-processing paths get the SAME head tensors
        cached = model.heads(x)
        model.heads = lambda _x: cached
        model.fused_graph = False          # this test pins the post-processing on GIVEN head tensors (eager graph)
        assert cached[0][0].dtype == torch.bfloat16 and not cached[0][0].is_contiguous()
        model.fused_postprocess = True
        fused = model(x)
        model.fused_postprocess = False
        plain = model(x)
    assert int((fused[0] > 0).sum()) > 50
    for f, p in zip(fused, plain):
        assert torch.equal(f, p)


@pytest.mark.parametrize('shape', [(2, 64, 256, 40, 56), (3, 256, 64, 20, 28), (1, 512, 2048, 7, 10), (2, 40, 24, 5, 3)])
@pytest.mark.parametrize('with_residual', [False, True])
@pytest.mark.parametrize('relu', [False, True])
def test_gemm_bias_act_matches_conv1x1(shape, with_residual, relu):
    """odtk_gemm_bias_act == conv1x1 + bias (+ residual) (+ ReLU) computed in float32 from the same bf16
    inputs, to bf16 rounding of the result (the GEMM accumulates in fp32 and rounds once)."""
    import torch.nn.functional as F
    b, ci, co, h, w = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(b, ci, h, w, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 1, 1, generator=g) * ci ** -0.5).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(co, generator=g).cuda()
    res = torch.randn(b, co, h, w, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    y = _C.gemm_bias_act(x, wt, bias, res if with_residual else None, relu)
    assert y.shape == (b, co, h, w) and y.is_contiguous(memory_format=torch.channels_last) and y.dtype == torch.bfloat16
    ref = F.conv2d(x.float(), wt.float()) + bias.view(1, -1, 1, 1)
    if with_residual:
        ref = ref + res.float()
    if relu:
        ref = ref.relu()
    err = (y.float() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 1e-2          # one bf16 rounding + fp32 accumulation order
    assert (err <= tol).all(), float((err - tol).max())
    # second call (plan cache hit) is bit-identical
    assert torch.equal(y, _C.gemm_bias_act(x, wt, bias, res if with_residual else None, relu))


@pytest.mark.parametrize('shape', [(2, 64, 50, 70), (1, 8, 7, 9), (3, 32, 16, 16), (1, 16, 1, 1)])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_bias_act_maxpool_is_epilogue_then_pool(shape, dtype):
    """The stem's fused bias + ReLU + 3x3/s2 max-pool equals bias_act_ followed by torch's max_pool2d
    bit for bit (monotone epilogue commutes with the maximum), odd sizes and borders included."""
    import torch.nn.functional as F
    b, c, h, w = shape
    g = torch.Generator().manual_seed(9)
    y = torch.randn(b, c, h, w, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(c, generator=g).cuda()
    for relu in (True, False):
        got = _C.bias_act_maxpool(y, bias, relu)
        ref = F.max_pool2d(_C.bias_act_(y.clone(memory_format=torch.channels_last), bias, None, relu), 3, 2, 1)
        assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(got, ref)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_bias_act_maxpool_special_values(dtype):
    """Round 6: the pool takes its maximum on order-preserving 16-bit keys of the raw patterns (csrc/epilogue.hpp
    order_keys16).  What the float form did by comparing must still hold: NaNs of EITHER sign propagate (as torch's
    max_pool2d), +-inf order as numbers, windows over the borders see clamped coordinates only.  Every finite position is bit
    for bit the reference; NaN positions are NaN in both."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(21)
    b, c, h, w = 2, 16, 23, 31
    y = torch.randn(b, c, h, w, generator=g).to(dtype)
    flat = y.view(-1)
    bits = flat.view(torch.int16)
    n = flat.numel()
    pick = torch.randperm(n, generator=g)
    flat[pick[:60]] = float('inf')
    flat[pick[60:160]] = float('-inf')
    flat[pick[160:200]] = float('nan')
    nan_bits = {torch.bfloat16: 0x7fc1, torch.float16: 0x7e01}[dtype]
    bits[pick[200:240]] = (nan_bits | 0x8000) - 0x10000                                      # NEGATIVE NaNs (sign bit set), a payload
    bits[pick[240:260]] = nan_bits | 0x3f                                        # positive NaNs with another payload
    flat[pick[260:400]] = 0.0
    bits[pick[400:520]] = -0x8000                                                # -0.0
    y[0, :, :3, :] = float('-inf')                                               # whole windows of -inf at a border
    y = y.cuda().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(c, generator=g).cuda()
    for relu in (True, False):
        got = _C.bias_act_maxpool(y, bias, relu)
        ref = F.max_pool2d(_C.bias_act_(y.clone(memory_format=torch.channels_last), bias, None, relu), 3, 2, 1)
        # ... and torch's own operators: relu and max_pool2d hand a NaN on (odtk_bias_act's ReLU turned it into 0 until round 6)
        t = y.float() + bias.view(1, -1, 1, 1)
        ref_torch = F.max_pool2d((torch.relu(t) if relu else t).to(dtype), 3, 2, 1)
        assert torch.equal(torch.isnan(ref), torch.isnan(ref_torch))
        assert torch.equal(ref[~torch.isnan(ref)].view(torch.int16), ref_torch[~torch.isnan(ref)].view(torch.int16))
        assert torch.equal(torch.isnan(got), torch.isnan(ref))
        assert int(torch.isnan(ref).sum()) > 100
        fin = ~torch.isnan(ref)
        assert torch.equal(got[fin].view(torch.int16), ref[fin].view(torch.int16))
        assert bool(torch.isinf(ref).any())


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('rotated', [False, True], ids=['axis', 'rotated'])
def test_head_bias_folded_into_the_kernels(dtype, rotated):
    """decode_levels(raw heads, cls_bias, box_bias) == the strict op on torch-materialised inputs:
    scores = sigmoid(float(raw) + bias) rounded to the head dtype, deltas = float(raw) + bias.
    Biases spread over several units so that per-channel thresholds matter; indices bit-exact."""
    a, c = (27, 8) if rotated else (9, 16)
    nb = 6 if rotated else 4
    g = torch.Generator().manual_seed(77)
    shapes, strides = [(37, 53), (19, 27), (5, 7)], [8, 16, 32]
    cls_bias = (torch.randn(a * c, generator=g) * 1.5 - 3.0).cuda()
    box_bias = (torch.randn(a * nb, generator=g) * 0.3).cuda()
    cls, box_h = [], []
    for h, w in shapes:
        cls.append((torch.randn(2, a * c, h, w, generator=g) * 1.2).to(dtype).cuda().contiguous(memory_format=torch.channels_last))
        box_h.append((torch.randn(2, a * nb, h, w, generator=g) * 0.3).to(dtype).cuda().contiguous(memory_format=torch.channels_last))
    if rotated:
        import math
        anchors = {s: box.generate_anchors_rotated(s, [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)],
                                                   [-math.pi / 6, 0, math.pi / 6])[0] for s in strides}
    else:
        anchors = {s: box.generate_anchors(s, [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)]) for s in strides}
    alist = [anchors[s] for s in strides]
    got = _C.decode_levels(cls, box_h, alist, strides, 0.05, 300, rotated, return_indices=True, logits=True,
                           cls_bias=cls_bias, box_bias=box_bias)
    scores = [(x.float() + cls_bias.view(1, -1, 1, 1)).sigmoid().to(dtype).float().contiguous() for x in cls]
    deltas = [(x.float() + box_bias.view(1, -1, 1, 1)).contiguous() for x in box_h]
    ref = _C.decode_levels(scores, deltas, alist, strides, 0.05, 300, rotated, return_indices=True)
    assert torch.equal(got[3], ref[3]), 'indices'
    assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2])
    assert torch.equal(got[1], ref[1])                    # same fp32 deltas into the same box arithmetic
    assert (got[0] > 0).sum().item() > 300                # the case is not vacuous
    # the prefilter's precomputed threshold table: the same outputs bit for bit; a table made for another threshold, dtype
    # or bias length is recognised by its key word and not used (the prefilter then passes everything to the exact test)
    table = _C.prefilter_thresholds(cls_bias, dtype, 0.05)
    assert table.numel() == a * c + 8
    other = torch.float16 if dtype == torch.bfloat16 else torch.bfloat16
    stale = [_C.prefilter_thresholds(cls_bias, dtype, 0.3), _C.prefilter_thresholds(cls_bias, other, 0.05),
             torch.zeros_like(table)]
    for t in [table] + stale:
        again = _C.decode_levels(cls, box_h, alist, strides, 0.05, 300, rotated, return_indices=True, logits=True,
                                 cls_bias=cls_bias, box_bias=box_bias, cls_thresholds=t)
        assert all(torch.equal(x, y) for x, y in zip(again, got))
    with pytest.raises(RuntimeError):                     # wrong length
        _C.decode_levels(cls, box_h, alist, strides, 0.05, 300, rotated, logits=True, cls_bias=cls_bias,
                         cls_thresholds=table[:-8].contiguous())
    # the fold is refused where it cannot be exact-by-construction
    with pytest.raises(RuntimeError):
        _C.decode_levels([x.float() for x in cls], [x.float() for x in box_h], alist, strides, 0.05, 300, rotated,
                         logits=True, cls_bias=cls_bias)
