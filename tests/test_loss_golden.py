"""SURVEY 8(f3) pinned to the REFERENCE's OWN loss modules (the fused kernel used to be checked only
against this repo's restatement of them).  tests/golden/loss_ref_*.npz come from /root/reference/odtk/loss.py, loaded from
where it lies and combined as reference odtk/model.py:193-209 does (oracle/gen_golden_loss.py), on the reference-generated
target fixtures (axis-aligned and rotated).

  * CPU (here): the product's torch modules (odtk/loss.py FocalLoss / SmoothL1Loss, what Model._compute_loss uses on the CPU
    and what the GPU tests used as their yardstick) reproduce the reference's sums and gradients -- the same torch
    expressions, so to float32 rounding; and, where /root/reference exists, element by element on random inputs.
  * GPU (tests/test_gpu_loss.py::test_fused_loss_against_reference_fixtures): the HIP kernel against the same files."""
import glob
import importlib.util
import os

import numpy as np
import pytest
import torch

from odtk import loss as L

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(glob.glob(os.path.join(GOLDEN, 'loss_ref_*.npz')))
REF_LOSS = '/root/reference/odtk/loss.py'


def load_case(path):
    with np.load(path) as z:
        g = {k: z[k] for k in z.files}
    with np.load(os.path.join(GOLDEN, str(g['targets']))) as z:
        t = {k: torch.from_numpy(z[k]).unsqueeze(0) for k in ('cls_target', 'box_target', 'depth')}
    return g, t


def product_level_loss(cls_head, box_head, cls_target, box_target, depth):
    """The CPU branch of Model._compute_loss for one level (odtk/model.py), on the product's own criteria."""
    cls_loss = L.FocalLoss()(cls_head.view_as(cls_target).float(), cls_target)
    box_loss = L.SmoothL1Loss(beta=0.11)(box_head.view_as(box_target).float(), box_target)
    return ((cls_loss * (depth >= 0).expand_as(cls_target).float()).sum(),
            (box_loss * (depth > 0).expand_as(box_target).float()).sum(), (depth > 0).sum())


def test_fixture_set_is_complete():
    assert len(CASES) == 7 and any('rot_' in c for c in CASES)


@pytest.mark.parametrize('path', CASES, ids=os.path.basename)
def test_product_torch_loss_reproduces_the_reference(path):
    g, t = load_case(path)
    cls_head = torch.from_numpy(g['cls_head']).requires_grad_(True)
    box_head = torch.from_numpy(g['box_head']).requires_grad_(True)
    c, b, fg = product_level_loss(cls_head, box_head, t['cls_target'], t['box_target'], t['depth'])
    assert float(fg) == float(g['sums32'][2])
    for got, k in ((c, 0), (b, 1)):
        want32, want64 = float(g['sums32'][k]), float(g['sums64'][k])
        got = float(got.detach())
        assert abs(got - want32) <= 2e-7 * abs(want32) + 1e-30, (k, got, want32)     # same expressions: fp32 rounding
        assert abs(got - want64) <= 3e-6 * abs(want64) + 1e-30
    (c * float(g['g'][0]) + b * float(g['g'][1])).backward()
    for mine, ref in ((cls_head.grad, g['dcls']), (box_head.grad, g['dbox'])):
        ref = torch.from_numpy(ref)
        assert float((mine - ref).abs().max()) <= 1e-6 * max(float(ref.abs().max()), 1e-30)


@pytest.mark.skipif(not os.path.isfile(REF_LOSS), reason='reference tree not present')
def test_product_torch_loss_equals_the_reference_elementwise():
    spec = importlib.util.spec_from_file_location('reference_odtk_loss', REF_LOSS)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 30, 9, 11, generator=g) * 3
    t = (torch.rand(4, 30, 9, 11, generator=g) < 0.1).float()
    for gamma in (2, 0.5, 0):
        assert torch.equal(L.FocalLoss(0.25, gamma)(x, t), ref.FocalLoss(0.25, gamma)(x, t))
    d = torch.randn(4, 36, 9, 11, generator=g) * 0.3
    assert torch.equal(L.SmoothL1Loss(0.11)(d, torch.zeros_like(d)), ref.SmoothL1Loss(0.11)(d, torch.zeros_like(d)))
