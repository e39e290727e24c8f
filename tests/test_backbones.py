"""Backbone architectures without torchvision: parameter counts are the published torchvision totals (any wrong width,
group count or missing layer changes them), state_dict keys follow torchvision's layout, features come out at strides
8 / 16 / 32 with the channel counts the FPN laterals expect."""
import pytest
import torch

from odtk import backbones
from odtk.model import Model

# total parameters of the torchvision classification models the reference subclasses (fc / classifier included)
PUBLISHED = {'ResNet18FPN': 11689512, 'ResNet34FPN': 21797672, 'ResNet50FPN': 25557032, 'ResNet101FPN': 44549160,
             'ResNet152FPN': 60192808, 'ResNeXt50_32x4dFPN': 25028904, 'ResNeXt101_32x8dFPN': 88791336,
             'MobileNetV2FPN': 3504872}


@pytest.mark.parametrize('name', sorted(PUBLISHED))
def test_parameter_count_equals_the_published_model(name):
    fpn = getattr(backbones, name)()
    assert sum(p.numel() for p in fpn.features.parameters()) == PUBLISHED[name]


@pytest.mark.parametrize('name,channels', [('ResNet18FPN', (128, 256, 512)), ('ResNeXt50_32x4dFPN', (512, 1024, 2048)),
                                           ('MobileNetV2FPN', (32, 96, 320))])
def test_features_and_pyramid_shapes(name, channels):
    fpn = getattr(backbones, name)().eval()
    x = torch.randn(1, 3, 128, 256)
    with torch.no_grad():
        feats = fpn.features(x)
        pyramid = fpn(x)
    assert [f.shape[1] for f in feats] == list(channels)
    assert [tuple(f.shape[2:]) for f in feats] == [(16, 32), (8, 16), (4, 8)]
    assert [tuple(p.shape[1:]) for p in pyramid] == [(256, 16, 32), (256, 8, 16), (256, 4, 8), (256, 2, 4), (256, 1, 2)]


def test_state_dict_keys_follow_torchvision():
    mobile = backbones.MobileNetV2FPN().features.state_dict()
    for key, shape in (('features.0.0.weight', (32, 3, 3, 3)), ('features.0.1.running_var', (32,)),
                       ('features.1.conv.0.0.weight', (32, 1, 3, 3)), ('features.1.conv.1.weight', (16, 32, 1, 1)),
                       ('features.1.conv.2.weight', (16,)), ('features.2.conv.0.0.weight', (96, 16, 1, 1)),
                       ('features.2.conv.1.0.weight', (96, 1, 3, 3)), ('features.2.conv.2.weight', (24, 96, 1, 1)),
                       ('features.2.conv.3.bias', (24,)), ('features.17.conv.2.weight', (320, 960, 1, 1)),
                       ('features.18.0.weight', (1280, 320, 1, 1)), ('classifier.1.weight', (1000, 1280))):
        assert tuple(mobile[key].shape) == shape, key
    assert len([k for k in mobile if k.endswith('num_batches_tracked')]) == 52
    resnext = backbones.ResNeXt101_32x8dFPN().features.state_dict()
    for key, shape in (('layer1.0.conv1.weight', (256, 64, 1, 1)), ('layer1.0.conv2.weight', (256, 8, 3, 3)),
                       ('layer1.0.conv3.weight', (256, 256, 1, 1)), ('layer1.0.downsample.0.weight', (256, 64, 1, 1)),
                       ('layer4.2.conv2.weight', (2048, 64, 3, 3)), ('fc.weight', (1000, 2048))):
        assert tuple(resnext[key].shape) == shape, key


@pytest.mark.parametrize('name', ['MobileNetV2FPN', 'ResNeXt50_32x4dFPN'])
def test_model_runs_and_round_trips_a_checkpoint(name, tmp_path):
    torch.manual_seed(0)
    model = Model(name, classes=4)
    model.initialize(None)
    assert model.stride == 128 and any(u in ('classifier', 'fc') for u in model.unused_modules)
    model.freeze_unused_params()
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    assert frozen and all(any(u in n for u in model.unused_modules) for n in frozen)
    with torch.no_grad():
        model.cls_head[-1].bias.fill_(0.0)
    x = torch.randn(1, 3, 128, 128)
    scores, boxes, classes = model.eval()(x)
    assert scores.shape == (1, 100) and boxes.shape == (1, 100, 4) and int((scores > 0).sum()) > 0
    path = str(tmp_path / 'm.pth')
    model.save({'path': path})
    again, _ = Model.load(path)
    out = again.eval()(x)
    assert torch.equal(out[0], scores) and torch.equal(out[1], boxes)
    loss = model.train()([x, torch.tensor([[[10., 10., 60., 60., 1.]]])])
    assert all(torch.isfinite(v) for v in loss)


def test_two_backbones_give_ten_levels_on_the_cpu_path():
    """`--backbone A B` (reference main.py:37, model.py:138): the heads run on both pyramids -- ten levels."""
    torch.manual_seed(0)
    model = Model(['ResNet18FPN', 'MobileNetV2FPN'], classes=3)
    model.initialize(None)
    with torch.no_grad():
        model.cls_head[-1].bias.fill_(0.0)
    x = torch.randn(1, 3, 128, 128)
    cls_heads, box_heads = model.eval().heads(x)
    assert len(cls_heads) == 10 and [c.shape[-1] for c in cls_heads] == [16, 8, 4, 2, 1] * 2
    scores, boxes, classes = model(x)
    assert scores.shape == (1, 100) and int((scores > 0).sum()) > 0
    loss = model.train()([x, torch.tensor([[[10., 10., 60., 60., 1.]]])])
    assert all(torch.isfinite(v) for v in loss)
    assert sorted(model.unused_modules) == ['classifier', 'fc', 'features.18']
