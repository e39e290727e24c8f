"""Feature Pyramid Network P3..P7 over a ResNet (https://arxiv.org/abs/1612.03144); module names
(lateral3-5, pyramid6-7, smooth3-5, features) follow reference odtk/backbones/fpn.py:12-61 so
checkpoints are interchangeable."""
import torch.nn as nn
import torch.nn.functional as F

from .resnet import ResNet, BasicBlock, Bottleneck


class FPN(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.stride = 128
        self.features = features
        c3, c4, c5 = (128, 256, 512) if features.bottleneck is BasicBlock else (512, 1024, 2048)
        self.lateral3 = nn.Conv2d(c3, 256, 1)
        self.lateral4 = nn.Conv2d(c4, 256, 1)
        self.lateral5 = nn.Conv2d(c5, 256, 1)
        self.pyramid6 = nn.Conv2d(c5, 256, 3, stride=2, padding=1)
        self.pyramid7 = nn.Conv2d(256, 256, 3, stride=2, padding=1)
        self.smooth3 = nn.Conv2d(256, 256, 3, padding=1)
        self.smooth4 = nn.Conv2d(256, 256, 3, padding=1)
        self.smooth5 = nn.Conv2d(256, 256, 3, padding=1)

    def initialize(self):
        for m in (self.lateral3, self.lateral4, self.lateral5, self.pyramid6, self.pyramid7,
                  self.smooth3, self.smooth4, self.smooth5):
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)
        self.features.initialize()

    def forward(self, x):
        c3, c4, c5 = self.features(x)
        p5 = self.lateral5(c5)
        p4 = self.lateral4(c4) + F.interpolate(p5, scale_factor=2)
        p3 = self.lateral3(c3) + F.interpolate(p4, scale_factor=2)
        p6 = self.pyramid6(c5)
        p7 = self.pyramid7(F.relu(p6))
        return [self.smooth3(p3), self.smooth4(p4), self.smooth5(p5), p6, p7]


def _fpn(layers, block):
    return FPN(ResNet(layers=layers, bottleneck=block, outputs=[3, 4, 5]))


def ResNet18FPN():
    return _fpn([2, 2, 2, 2], BasicBlock)


def ResNet34FPN():
    return _fpn([3, 4, 6, 3], BasicBlock)


def ResNet50FPN():
    return _fpn([3, 4, 6, 3], Bottleneck)


def ResNet101FPN():
    return _fpn([3, 4, 23, 3], Bottleneck)


def ResNet152FPN():
    return _fpn([3, 8, 36, 3], Bottleneck)
