"""libodtk_conv.so (include/odtk_conv.h, csrc/conv_ck.cpp): the engine's k x k convolution with bias + ReLU in the convolution's
own epilogue, against a plain PyTorch fp32 reference of the same op, and the engine's per-layer plan against round 4's graph."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]

pytestmark = pytest.mark.gpu


def _reference(x, w, b, stride, pad, relu):
    """Plain PyTorch fp32 reference of the op, on the CPU: the GPU's own fp32 convolution is not a reference at this tolerance
    (MIOpen may pick a Winograd / reduced-precision solver for fp32; the first run of this test measured it 1 bf16 ulp away)."""
    y = F.conv2d(x.float().cpu().contiguous(), w.float().cpu().contiguous(), b.float().cpu(), stride, pad)
    return (F.relu(y) if relu else y).to(x.device)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [
    # batch, c_in, h, w, c_out, k, stride, pad, relu
    (2, 256, 25, 40, 256, 3, 1, 1, True),        # head tower, P5
    (2, 256, 50, 80, 256, 3, 1, 1, False),       # FPN smoothing convolution
    (2, 64, 56, 72, 64, 3, 1, 1, True),          # layer1 conv2
    (2, 128, 56, 72, 128, 3, 2, 1, True),        # layer2 conv2 of the first block (stride 2)
    (1, 2048, 13, 20, 256, 3, 2, 1, False),      # pyramid6
    (2, 256, 7, 10, 256, 3, 2, 1, False),        # pyramid7 (odd extents)
])
def test_conv_bias_act_matches_the_fp32_reference(dtype, shape):
    from odtk import _C
    assert _C.conv_available(), 'libodtk_conv.so is missing on a GPU box: build it (make -C retinanet-examples_amd/csrc conv)'
    b, c, h, w, k, ks, stride, pad, relu = shape
    g = torch.Generator().manual_seed(sum(shape[:5]))
    x = (torch.randn(b, c, h, w, generator=g) * 0.5).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(k, c, ks, ks, generator=g) * (2.0 / (c * ks * ks)) ** 0.5).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    bias = (torch.randn(k, generator=g) * 0.3).to(dtype).cuda()
    y = _C.conv_bias_act(x, wt, bias, stride, pad, relu)
    torch.cuda.synchronize()
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    ref = _reference(x, wt, bias, stride, pad, relu)
    assert y.shape == ref.shape
    # fp32 accumulation, but TWO roundings to the 16-bit type: the instance lists shuffle the accumulator through LDS in the
    # activation type before the epilogue adds the bias and rounds again (CShuffleDataType = the output type) -- the same two
    # roundings the two-launch form makes (convolution output, then odtk_bias_act): one ulp of the result in all
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    err = (y.float() - ref).abs()
    tol = eps * ref.abs() + eps * bias.float().abs().max() + 1e-3
    worst = int((err - tol).argmax())
    assert bool((err <= tol).all()), 'excess %.3g at ref %.4g, got %.4g (%s)' % (
        float((err - tol).flatten()[worst]), float(ref.flatten()[worst]), float(y.float().flatten()[worst]), _C.conv_last_plan())
    if relu:
        assert float(y.float().min()) >= 0.0
    # the second call only enqueues and gives the same bits
    y2 = _C.conv_bias_act(x, wt, bias, stride, pad, relu)
    assert torch.equal(y, y2)
    assert _C.conv_last_plan().startswith('#')


def test_unsupported_problems_raise_instead_of_computing_something_else():
    from odtk import _C
    x = torch.randn(1, 3, 32, 32).bfloat16().cuda().contiguous(memory_format=torch.channels_last)     # c_in = 3: no vector width fits
    w = torch.randn(64, 3, 7, 7).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.zeros(64).bfloat16().cuda()
    try:
        y = _C.conv_bias_act(x, w, b, 2, 3, True)
    except RuntimeError as e:
        assert 'unsupported' in str(e).lower() or 'conv_bias_act' in str(e)
    else:                                                   # an instance took it after all: then it must be right
        ref = _reference(x, w, b, 2, 3, True)
        assert torch.allclose(y.float(), ref, rtol=2 ** -7, atol=1e-2)
    with pytest.raises(RuntimeError):
        _C.conv_bias_act(x.float(), w.float(), b.float(), 2, 3, True)       # fp32: not offered


def test_engine_plan_keeps_the_detections():
    """The engine with its per-layer plan (some k x k convolutions through the library's fused epilogue) against the same engine
    with every k x k convolution as MIOpen + odtk_bias_act: same head tensors up to the rounding the epilogue no longer does
    twice, and the plan covers every k x k convolution that has an epilogue."""
    from odtk import fused
    from odtk.model import Model
    torch.manual_seed(0)
    model = Model('ResNet18FPN', classes=20).cuda().eval()
    model.initialize(None)
    x = torch.randn(2, 3, 256, 320, device='cuda')
    fused._Conv.use_conv_library = False
    try:
        e0 = fused.FusedRetinaNet(model, torch.bfloat16)
        with torch.no_grad():
            c0, b0 = e0.heads(x)
    finally:
        fused._Conv.use_conv_library = True
    e1 = fused.FusedRetinaNet(model, torch.bfloat16)
    with torch.no_grad():
        e1.plan(x)
        routes = e1.conv_routes()
        assert routes and all(len(v) >= 1 for v in routes.values())
        c1, b1 = e1.heads(x)
        # force every measured layer through the library, whatever the A/B said: the numerical comparison must not depend on timing
        for mod in e1.modules():
            if isinstance(mod, fused._Conv):
                mod.route = {k: (v[1] != float('inf'), v[1], v[2]) for k, v in mod.route.items()}
        c2, b2 = e1.heads(x)
    for a, b in zip(c0 + b0, c2 + b2):
        scale = float(a.float().abs().max())
        assert float((a.float() - b.float()).abs().max()) <= 0.03 * scale + 1e-3, 'head tensors drifted'
    for a, b in zip(c0 + b0, c1 + b1):
        scale = float(a.float().abs().max())
        assert float((a.float() - b.float()).abs().max()) <= 0.03 * scale + 1e-3


@pytest.mark.parametrize('in_dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
def test_stem_pack_is_space_to_depth_plus_cast(in_dtype, layout):
    from odtk import _C
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 64, 96, generator=g).to(in_dtype).cuda()
    if layout == 'channels_last':
        x = x.contiguous(memory_format=torch.channels_last)
    for dtype in (torch.bfloat16, torch.float16):
        got = _C.stem_pack(x, dtype)
        b, _, h, w = x.shape
        ref = x.float().view(b, 3, h // 2, 2, w // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(b, 12, h // 2, w // 2)
        ref = torch.cat([ref, torch.zeros(b, 4, h // 2, w // 2, device=x.device)], 1).to(dtype)
        assert got.shape == ref.shape and got.dtype == dtype and got.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(got, ref.contiguous(memory_format=torch.channels_last))       # a cast and a permutation: bit for bit


def test_stem_in_space_to_depth_form_equals_the_direct_stem():
    """conv7x7/s2/p3 over 3 channels == conv4x4/s1/pad(2, 1) over the packed input with the re-indexed weights: the same products in
    another order (fp32 accumulation, one rounding to bf16 of the same sums -> equal up to an ulp where the order matters)."""
    from odtk import fused
    from odtk.model import Model
    torch.manual_seed(0)
    model = Model('ResNet18FPN', classes=4).cuda().eval()
    model.initialize(None)
    e = fused.FusedRetinaNet(model, torch.bfloat16)
    assert e.stem_s2d is not None
    x = torch.randn(2, 3, 128, 192, device='cuda')
    with torch.no_grad():
        direct = e._stem_direct(x)
        packed = e._stem_packed(x)
    assert direct.shape == packed.shape
    diff = (direct.float() - packed.float()).abs()
    scale = float(direct.float().abs().max())
    assert float(diff.max()) <= 2.0 ** -7 * scale + 1e-3, (float(diff.max()), scale)
    # (about half of the outputs differ by one bf16 ulp: the two convolutions sum the same products in different orders)
    # channels_last input and an fp16 engine take the same route
    e16 = fused.FusedRetinaNet(model, torch.float16)
    xc = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        d16, p16 = e16._stem_direct(xc), e16._stem_packed(xc)
    assert float((d16.float() - p16.float()).abs().max()) <= 2.0 ** -10 * float(d16.float().abs().max()) + 1e-3


def test_many_input_geometries_plan_a_few_and_follow_the_rest():
    """A data set's batches are padded to the largest image of the batch: dozens of geometries.  The engine plans the first
    `max_plans`, later ones follow the layers' last measured decisions (and the library adopts a sibling problem's instance
    instead of tuning again); the head tensors stay those of the two-launch graph either way."""
    from odtk import fused
    from odtk.model import Model
    torch.manual_seed(0)
    model = Model('ResNet18FPN', classes=6).cuda().eval()
    model.initialize(None)
    sizes = [(256, 320), (256, 384), (320, 320), (192, 256), (320, 384), (256, 256), (384, 384)]
    e = fused.FusedRetinaNet(model, torch.bfloat16)
    e.max_plans = 3
    xs = [torch.randn(2, 3, h, w, device='cuda') for h, w in sizes]
    outs = []
    with torch.no_grad():
        for x in xs:
            e.plan(x)                                                   # (what forward() does first)
            outs.append(e.heads(x))
    assert len(e._planned) == 3                                         # the rest ran without a plan pass
    fused._Conv.use_conv_library = False
    try:
        ref_engine = fused.FusedRetinaNet(model, torch.bfloat16)
        with torch.no_grad():
            refs = [ref_engine.heads(x) for x in xs]
    finally:
        fused._Conv.use_conv_library = True
    for (c, b), (rc, rb) in zip(outs, refs):
        for got, ref in zip(c + b, rc + rb):
            scale = float(ref.float().abs().max())
            assert float((got.float() - ref.float()).abs().max()) <= 0.03 * scale + 1e-3
