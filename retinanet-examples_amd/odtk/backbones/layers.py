"""Frozen batch norm (statistics and affine parameters are buffers), reference
odtk/backbones/layers.py:5-32."""
import torch
from torch import nn
import torch.nn.functional as F


class FixedBatchNorm2d(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.register_buffer('weight', torch.ones(n))
        self.register_buffer('bias', torch.zeros(n))
        self.register_buffer('running_mean', torch.zeros(n))
        self.register_buffer('running_var', torch.ones(n))

    def forward(self, x):
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias)


def convert_fixedbn_model(module):
    """Recursively swap nn.BatchNorm2d for FixedBatchNorm2d, keeping the tensors."""
    out = module
    if isinstance(module, nn.BatchNorm2d):
        out = FixedBatchNorm2d(module.num_features)
        out.running_mean = module.running_mean
        out.running_var = module.running_var
        if module.affine:
            out.weight.data = module.weight.data.clone().detach()
            out.bias.data = module.bias.data.clone().detach()
    for name, child in module.named_children():
        out.add_module(name, convert_fixedbn_model(child))
    return out
