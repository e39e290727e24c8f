"""Inference fusion (odtk/fused.py + csrc/epilogue.hpp): the HIP epilogue against torch, and the
BN-folded fused graph against the eager model."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from odtk import _C
from odtk.fused import FusedRetinaNet, fold_conv_bn
from odtk.model import Model


def test_fold_conv_bn_is_exact_in_fp32_cpu():
    torch.manual_seed(0)
    conv = nn.Conv2d(8, 12, 3, padding=1, bias=True)
    bn = nn.BatchNorm2d(12).eval()
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(2, 8, 9, 7)
    w, b = fold_conv_bn(conv, bn)
    ref = bn(conv(x))
    got = F.conv2d(x, w, None, 1, 1) + b.view(1, -1, 1, 1)
    assert torch.allclose(ref, got, atol=1e-5, rtol=1e-5)
    w2, b2 = fold_conv_bn(conv, None)
    assert torch.equal(w2, conv.weight) and torch.equal(b2, conv.bias)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32], ids=['bf16', 'fp16', 'fp32'])
@pytest.mark.parametrize('shape', [(2, 64, 17, 23), (1, 36, 7, 10), (3, 720, 5, 3), (2, 7, 3, 3), (1, 256, 1, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_bias_act_matches_torch(dtype, shape):
    g = torch.Generator().manual_seed(sum(shape))
    y = torch.randn(*shape, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    r = torch.randn(*shape, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(shape[1], generator=g).cuda()
    for res in (None, r):
        for relu in (False, True):
            ref = y.float() + bias.view(1, -1, 1, 1) + (res.float() if res is not None else 0)
            ref = (F.relu(ref) if relu else ref).to(dtype)
            out = _C.bias_act_(y.clone(memory_format=torch.preserve_format), bias, res, relu)
            assert out.dtype == dtype and out.shape == y.shape
            assert torch.equal(out, ref), (res is not None, relu)       # fp32 math, one rounding: exact


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32], ids=['bf16', 'fp16', 'fp32'])
@pytest.mark.parametrize('shape', [(2, 256, 25, 40), (1, 256, 13, 20), (3, 8, 1, 1), (2, 64, 7, 5)], ids=lambda s: 'x'.join(map(str, s)))
def test_upsample2x_matches_torch_nearest(dtype, shape):
    """The FPN's top-down upsampling (reference fpn.py:45-61) as one HIP stream kernel: bytes are copied, so it equals
    F.interpolate(scale_factor=2) bit for bit (NaN payloads included); unsupported layouts raise."""
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    x.permute(0, 2, 3, 1).reshape(-1)[::97] = float('nan')             # (a view: channels_last storage is NHWC-contiguous)
    out = _C.upsample2x(x)
    ref = F.interpolate(x, scale_factor=2)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out.view(torch.int16 if dtype != torch.float32 else torch.int32), ref.contiguous(memory_format=torch.channels_last).view(
        torch.int16 if dtype != torch.float32 else torch.int32))
    if shape[2] * shape[3] > 1:
        with pytest.raises(RuntimeError):
            _C.upsample2x(x.contiguous())                              # NCHW
    with pytest.raises(RuntimeError):
        _C.upsample2x(torch.zeros(1, 3, 4, 4, device='cuda', dtype=dtype).contiguous(memory_format=torch.channels_last))


@pytest.mark.gpu
@pytest.mark.parametrize('backbone', ['ResNet18FPN', 'ResNet50FPN'])
def test_fused_graph_matches_eager_fp32(backbone):
    torch.manual_seed(0)
    model = Model(backbone, classes=20)
    model.initialize(None)
    for m in model.modules():                       # non-trivial frozen statistics
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.8, 1.2); m.bias.data.normal_(0, 0.1)
    model = model.cuda().to(memory_format=torch.channels_last).eval()
    x = torch.randn(2, 3, 256, 320, device='cuda').contiguous(memory_format=torch.channels_last)
    fused = FusedRetinaNet(model, dtype=torch.float32)
    with torch.no_grad():
        ref_cls, ref_box = model.heads(x)
        got_cls, got_box = fused.heads(x)
    for r, g in zip(ref_cls + ref_box, got_cls + got_box):
        assert r.shape == g.shape
        scale = r.abs().max().item() + 1e-6
        assert (r - g).abs().max().item() <= 2e-3 * scale, (r - g).abs().max().item() / scale
    # bf16 fused graph: same function up to bf16 rounding; detections come out of the HIP path
    fused16 = FusedRetinaNet(model, dtype=torch.bfloat16)
    with torch.no_grad():
        c16, _ = fused16.heads(x)
        det = fused16(x)
    for r, g in zip(ref_cls, c16):
        cos = F.cosine_similarity(r.flatten().float(), g.flatten().float(), dim=0).item()
        assert cos > 0.999, cos
    assert det[0].shape == (2, 100) and det[1].shape == (2, 100, 4)


@pytest.mark.gpu
def test_engine_on_a_128px_image_whose_p7_level_is_1x1():
    """a [B, C, 1, 1] head is NCHW- and NHWC-contiguous at once; the bias fold needs the channels_last reading.
    Also the default route: Model.forward in eval mode IS the engine, under autocast and without."""
    torch.manual_seed(0)
    model = Model('ResNet18FPN', classes=8)
    model.initialize(None)
    model = model.cuda().to(memory_format=torch.channels_last).eval()
    x = torch.randn(2, 3, 128, 128, device='cuda')
    with torch.no_grad():
        cls_heads, _ = model.heads(x)
        assert cls_heads[-1].shape[-2:] == (1, 1)
        model.cls_head[-1].weight.mul_(60.0)                  # detections exist
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out16 = model(x)                                    # bf16 engine, head bias folded into the kernels
        assert torch.bfloat16 in model._engine_cache
        out32 = model(x)                                        # fp32 engine (cache re-keyed by dtype)
        assert torch.float32 in model._engine_cache and torch.bfloat16 not in model._engine_cache
        model.fused_graph = False
        eager = model(x)
    for o in (out16, out32, eager):
        assert o[0].shape == (2, 100) and o[1].shape == (2, 100, 4)
    assert int((out32[0] > 0).sum()) > 10
    # fp32 engine vs fp32 eager graph: same detector up to conv rounding
    n = int(min((out32[0] > 0).sum(), (eager[0] > 0).sum()))
    assert (out32[0].flatten().sort(descending=True).values[:n // 2] - eager[0].flatten().sort(descending=True).values[:n // 2]).abs().max() < 1e-3
    # the engine follows the weights: an in-place update re-folds it
    model.fused_graph = True
    with torch.no_grad():
        before = model(x)[0].clone()
        model.cls_head[-1].bias.add_(0.5)
        after = model(x)[0]
    assert not torch.equal(before, after)


def test_engine_cache_follows_the_weights_cpu():
    """what the cached engine notices (no GPU needed: engines are only built here, never run).
    In-place updates (optimizer steps, load_state_dict) bump the version counter -> re-fold on the next call.  A REPLACED
    Parameter / submodule keeps the old tensor objects alive in the cache with unchanged versions: caught by the module walk
    that follows every train() / eval() switch, or by invalidate_engine(); writes through `.data` are invisible to both and
    need invalidate_engine() (documented on the method)."""
    torch.manual_seed(0)
    model = Model('ResNet18FPN', classes=4)
    model.initialize(None)
    model.eval()
    e1 = model.inference_engine(torch.float32)
    assert model.inference_engine(torch.float32) is e1                       # unchanged weights: cache hit
    with torch.no_grad():
        model.cls_head[0].weight.mul_(1.5)                                   # in-place: version bump
    e2 = model.inference_engine(torch.float32)
    assert e2 is not e1
    # a replaced layer, between two eval calls with no mode switch: NOT seen (documented) ...
    old = model.box_head[0]
    model.box_head[0] = nn.Conv2d(256, 256, 3, padding=1)
    assert model.inference_engine(torch.float32) is e2
    # ... seen after the next train() / eval() switch (what every training loop does before validating)
    model.train()
    model.eval()
    e3 = model.inference_engine(torch.float32)
    assert e3 is not e2
    assert torch.equal(e3.box_head[0].weight.float(), model.box_head[0].weight.detach())
    assert not torch.equal(e3.box_head[0].weight.float(), old.weight.detach())
    assert model.inference_engine(torch.float32) is e3                       # the walk is not repeated on every call
    # a replaced Parameter object
    model.cls_head[2].weight = nn.Parameter(torch.zeros_like(model.cls_head[2].weight))
    model.eval()
    e4 = model.inference_engine(torch.float32)
    assert e4 is not e3 and float(e4.cls_head[1].weight.abs().max()) == 0.0
    # .data writes: invisible, the public hook drops the engine
    model.cls_head[2].weight.data.fill_(0.25)
    assert model.inference_engine(torch.float32) is e4
    model.invalidate_engine()
    e5 = model.inference_engine(torch.float32)
    assert e5 is not e4 and float(e5.cls_head[1].weight.float().mean()) == 0.25
