"""ResNet feature extractor (https://arxiv.org/abs/1512.03385) without torchvision.

Module / parameter names follow torchvision.models.resnet (conv1, bn1, layer1..4, `downsample.0/.1`,
fc) because the reference's backbone IS a torchvision ResNet subclass (reference
odtk/backbones/resnet.py:8-22) and its checkpoints carry those keys.  `forward` returns the
feature maps of the requested levels (C3, C4, C5 for FPN; reference resnet.py:24-39).
"""
import torch.nn as nn


def _conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, width, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(cin, width, stride)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(width, width)
        self.bn2 = nn.BatchNorm2d(width)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + skip)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride=1, downsample=None, groups=1, base_width=64):
        super().__init__()
        out = width * 4
        width = int(width * (base_width / 64.0)) * groups     # ResNeXt: `groups` paths of `base_width`-scaled width
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, groups=groups, bias=False)   # stride on the 3x3 ("v1.5")
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, out, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + skip)


class ResNet(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), bottleneck=Bottleneck, outputs=(5,), groups=1, width_per_group=64):
        super().__init__()
        if (groups, width_per_group) != (1, 64) and bottleneck is not Bottleneck:
            raise ValueError('grouped / widened paths (ResNeXt) need the Bottleneck block')
        extra = {'groups': groups, 'base_width': width_per_group} if bottleneck is Bottleneck else {}
        self.stride = 128
        self.bottleneck = bottleneck
        self.outputs = list(outputs)
        self.unused_modules = ['fc']

        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = 64
        for i, (depth, width) in enumerate(zip(layers, (64, 128, 256, 512))):
            blocks = []
            for j in range(depth):
                stride = 2 if (j == 0 and i > 0) else 1
                cout = width * bottleneck.expansion
                down = None
                if stride != 1 or cin != cout:
                    down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))
                blocks.append(bottleneck(cin, width, stride, down, **extra))
                cin = cout
            setattr(self, 'layer%d' % (i + 1), nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(cin, 1000)       # kept for checkpoint-key compatibility; never used
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def initialize(self):
        """The reference downloads ImageNet weights here (resnet.py:20-22); there is no network on this
        platform, so the random initialisation stands (load a checkpoint with Model.load instead)."""

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        feats = []
        for level, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4), start=2):
            if level > max(self.outputs):
                break
            x = layer(x)
            if level in self.outputs:
                feats.append(x)
        return feats
