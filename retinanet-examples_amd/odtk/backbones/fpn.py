"""Feature Pyramid Network P3..P7 (https://arxiv.org/abs/1612.03144) on top of a ResNet, ResNeXt or MobileNetV2.

Attribute names (features, lateral3-5, pyramid6-7, smooth3-5) and therefore state_dict keys are the
ones reference checkpoints use (reference odtk/backbones/fpn.py:12-61)."""
import torch.nn as nn
import torch.nn.functional as F

from .mobilenet import MobileNet
from .resnet import ResNet, BasicBlock, Bottleneck

FPN_CHANNELS = 256
RESNET_DEPTHS = {'ResNet18FPN': ([2, 2, 2, 2], BasicBlock), 'ResNet34FPN': ([3, 4, 6, 3], BasicBlock),
                 'ResNet50FPN': ([3, 4, 6, 3], Bottleneck), 'ResNet101FPN': ([3, 4, 23, 3], Bottleneck),
                 'ResNet152FPN': ([3, 8, 36, 3], Bottleneck)}
# (depths, groups, width per group) -- reference fpn.py:83-89
RESNEXT = {'ResNeXt50_32x4dFPN': ([3, 4, 6, 3], 32, 4), 'ResNeXt101_32x8dFPN': ([3, 4, 23, 3], 32, 8)}


class FPN(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.stride = 128                      # P7
        self.features = features
        if isinstance(features, MobileNet):
            widths = [32, 96, 320]                                                       # features 6, 13, 17
        else:
            widths = [w * features.bottleneck.expansion for w in (128, 256, 512)]      # C3, C4, C5
        for level, width in zip((3, 4, 5), widths):
            setattr(self, 'lateral%d' % level, nn.Conv2d(width, FPN_CHANNELS, kernel_size=1))
        self.pyramid6 = nn.Conv2d(widths[-1], FPN_CHANNELS, kernel_size=3, stride=2, padding=1)
        self.pyramid7 = nn.Conv2d(FPN_CHANNELS, FPN_CHANNELS, kernel_size=3, stride=2, padding=1)
        for level in (3, 4, 5):
            setattr(self, 'smooth%d' % level, nn.Conv2d(FPN_CHANNELS, FPN_CHANNELS, kernel_size=3, padding=1))

    def own_convs(self):
        return [m for n, m in self.named_children() if n != 'features']

    def initialize(self):
        for conv in self.own_convs():
            nn.init.xavier_uniform_(conv.weight)
            nn.init.zeros_(conv.bias)
        self.features.initialize()

    def forward(self, x):
        c3, c4, c5 = self.features(x)
        top_down = self.lateral5(c5)
        pyramid = {5: top_down}
        for level, feat in ((4, c4), (3, c3)):                  # add the upsampled coarser map
            top_down = getattr(self, 'lateral%d' % level)(feat) + F.interpolate(top_down, scale_factor=2)
            pyramid[level] = top_down
        p6 = self.pyramid6(c5)
        p7 = self.pyramid7(F.relu(p6))
        return [self.smooth3(pyramid[3]), self.smooth4(pyramid[4]), self.smooth5(pyramid[5]), p6, p7]


def _make(name):
    layers, block = RESNET_DEPTHS[name]

    def build():
        return FPN(ResNet(layers=layers, bottleneck=block, outputs=[3, 4, 5]))
    build.__name__ = name
    return build


ResNet18FPN, ResNet34FPN, ResNet50FPN, ResNet101FPN, ResNet152FPN = (_make(n) for n in RESNET_DEPTHS)


def ResNeXt50_32x4dFPN():
    layers, groups, width = RESNEXT['ResNeXt50_32x4dFPN']
    return FPN(ResNet(layers=layers, bottleneck=Bottleneck, outputs=[3, 4, 5], groups=groups, width_per_group=width))


def ResNeXt101_32x8dFPN():
    layers, groups, width = RESNEXT['ResNeXt101_32x8dFPN']
    return FPN(ResNet(layers=layers, bottleneck=Bottleneck, outputs=[3, 4, 5], groups=groups, width_per_group=width))


def MobileNetV2FPN():
    return FPN(MobileNet(outputs=[6, 13, 17]))
