"""Frozen batch normalisation for fine-tuning / inference.

`FixedBatchNorm2d` keeps the four tensors of an nn.BatchNorm2d as BUFFERS (so they are neither
trained nor all-reduced by DDP -- reference odtk/backbones/layers.py:5-16 does the same) under the
same state_dict keys (weight, bias, running_mean, running_var), and applies them as one fused affine
map y = x * scale + shift with scale = weight / sqrt(var + eps), shift = bias - mean * scale.
`convert_fixedbn_model` swaps every nn.BatchNorm2d of a module tree in place (reference
layers.py:18-32); odtk.fused folds the same scale/shift into the convolution for inference."""
import torch
from torch import nn

EPS = 1e-5      # F.batch_norm's default, which the reference's FixedBatchNorm2d relies on


class FixedBatchNorm2d(nn.Module):
    def __init__(self, num_features, eps=EPS):
        super().__init__()
        self.eps = eps
        for name, fill in (('weight', 1.0), ('bias', 0.0), ('running_mean', 0.0), ('running_var', 1.0)):
            self.register_buffer(name, torch.full((num_features,), fill))

    def affine(self):
        scale = self.weight * torch.rsqrt(self.running_var + self.eps)
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        scale, shift = self.affine()
        return x * scale.to(x.dtype).view(1, -1, 1, 1) + shift.to(x.dtype).view(1, -1, 1, 1)

    @classmethod
    def from_batchnorm(cls, bn):
        frozen = cls(bn.num_features, bn.eps)
        frozen.running_mean, frozen.running_var = bn.running_mean, bn.running_var
        if bn.affine:
            frozen.weight, frozen.bias = bn.weight.detach().clone(), bn.bias.detach().clone()
        return frozen


def convert_fixedbn_model(module):
    """Returns `module` with every nn.BatchNorm2d (at any depth) replaced by a FixedBatchNorm2d that
    carries the same statistics and affine parameters."""
    if isinstance(module, nn.BatchNorm2d):
        return FixedBatchNorm2d.from_batchnorm(module)
    for name, child in list(module.named_children()):
        swapped = convert_fixedbn_model(child)
        if swapped is not child:
            setattr(module, name, swapped)
    return module
