"""The identity behind the engine's stem (odtk/fused.py, include/odtk_hip.h: odtk_stem_pack), in plain torch on the CPU:
conv7x7 / stride 2 / pad 3 over x  ==  conv4x4 / stride 1 / pad (2 before, 1 after) over the 2x2 space-to-depth image of x with
the re-indexed weights -- same products, another summation order (float64 here: equal to rounding)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]


@pytest.mark.parametrize('shape', [(2, 64, 64, 96), (1, 16, 38, 50), (3, 8, 2, 2)])
def test_stem_identity(shape):
    from odtk.fused import space_to_depth_pack, stem_space_to_depth_weights
    b, k, h, w = shape
    g = torch.Generator().manual_seed(h * w)
    x = torch.randn(b, 3, h, w, generator=g, dtype=torch.float64)
    wt = torch.randn(k, 3, 7, 7, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, wt, None, stride=2, padding=3)
    xs = F.pad(space_to_depth_pack(x), (2, 1, 2, 1))                   # (left, right, top, bottom) = 2 before, 1 after
    got = F.conv2d(xs, stem_space_to_depth_weights(wt), None, stride=1, padding=0)
    assert got.shape == ref.shape == (b, k, h // 2, w // 2)
    assert float((got - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
    # 147 of the 256 re-indexed taps carry a weight, the other 109 are structural zeros (the (0, 0) sub-pixel's first row /
    # column and the four padding channels)
    w4 = stem_space_to_depth_weights(torch.ones(1, 3, 7, 7))
    assert int(w4.sum()) == 147 and w4.shape == (1, 16, 4, 4)
