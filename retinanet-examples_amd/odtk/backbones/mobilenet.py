"""MobileNetV2 feature extractor (https://arxiv.org/abs/1801.04381) without torchvision.

Module layout and therefore state_dict keys are torchvision.models.mobilenet_v2's (`features.0.0.weight`,
`features.N.conv.K...`, `features.18.*`, `classifier.1.*`): the reference's backbone IS that class (reference
odtk/backbones/mobilenet.py:5-25) and its checkpoints carry those keys.  `forward` returns the outputs of the requested
feature indices (6, 13, 17 = strides 8, 16, 32 with 32, 96, 320 channels for the FPN); the last 1x1 convolution
(`features.18`) and the classifier exist for the keys only."""
import torch.nn as nn

# expansion t, channels c, repeats n, stride s (table 2 of the paper)
SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def conv_bn_relu6(cin, cout, kernel=3, stride=1, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel, stride, (kernel - 1) // 2, groups=groups, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU6(inplace=True))


class InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, expand):
        super().__init__()
        hidden = int(round(cin * expand))
        self.use_res_connect = stride == 1 and cin == cout
        layers = [conv_bn_relu6(cin, hidden, kernel=1)] if expand != 1 else []
        layers += [conv_bn_relu6(hidden, hidden, stride=stride, groups=hidden),          # depthwise
                   nn.Conv2d(hidden, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout)]  # linear projection
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


class MobileNet(nn.Module):
    def __init__(self, outputs=(18,), num_classes=1000):
        super().__init__()
        self.stride = 128
        self.outputs = list(outputs)
        self.unused_modules = ['features.18', 'classifier']
        cin, features = 32, [conv_bn_relu6(3, 32, stride=2)]
        for t, c, n, s in SETTING:
            for i in range(n):
                features.append(InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        features.append(conv_bn_relu6(cin, 1280, kernel=1))
        self.features = nn.Sequential(*features)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)

    def initialize(self):
        """The reference downloads ImageNet weights here (mobilenet.py:15-17); there is no network on this platform."""

    def forward(self, x):
        feats = []
        for index, layer in enumerate(self.features[:-1]):
            x = layer(x)
            if index in self.outputs:
                feats.append(x)
        return feats
