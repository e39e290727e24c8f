"""Backbones for the RetinaNet model: plain PyTorch-ROCm modules (MIOpen / hipBLASLt own the MFMA
work).  torchvision is not a dependency; state_dict keys match torchvision's ResNet so reference
checkpoints (reference odtk/backbones/resnet.py subclasses torchvision.models.resnet.ResNet) load."""
from .resnet import ResNet, BasicBlock, Bottleneck
from .fpn import (FPN, ResNet18FPN, ResNet34FPN, ResNet50FPN, ResNet101FPN, ResNet152FPN, ResNeXt50_32x4dFPN,
                  ResNeXt101_32x8dFPN, MobileNetV2FPN)
from .mobilenet import MobileNet
from .layers import FixedBatchNorm2d, convert_fixedbn_model

__all__ = ['ResNet', 'BasicBlock', 'Bottleneck', 'FPN', 'ResNet18FPN', 'ResNet34FPN', 'ResNet50FPN',
           'ResNet101FPN', 'ResNet152FPN', 'ResNeXt50_32x4dFPN', 'ResNeXt101_32x8dFPN', 'MobileNetV2FPN', 'MobileNet',
           'FixedBatchNorm2d', 'convert_fixedbn_model']
