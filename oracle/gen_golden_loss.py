#!/usr/bin/env python
"""Golden vectors for SURVEY 8(f3) -- the fused FocalLoss / SmoothL1 masked reduction -- from the REFERENCE's OWN modules.

TEST INFRASTRUCTURE ONLY (build container: needs /root/reference).  /root/reference/odtk/loss.py is pure torch and is loaded
from where it lies (importlib, nothing copied); it is combined exactly as reference odtk/model.py:193-209 combines it
(view_as, .float(), the two masks, the sums) on the reference-generated target fixtures tests/golden/snap_*.npz and
snaprot_ref_*.npz, for seeded head tensors.  Stored per case: the heads, the three sums in float32 (what the reference
computes) and float64 (the same modules on float64 inputs: the truth the kernel's 1e-6 bar is measured against), and the
autograd gradients of `g_cls * cls_sum + g_box * box_sum` with respect to both heads (float32).

    python oracle/gen_golden_loss.py        ->  tests/golden/loss_ref_*.npz
"""
import glob
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
REF_LOSS = '/root/reference/odtk/loss.py'
G_CLS, G_BOX = 0.37, -1.9


def reference_loss_module():
    spec = importlib.util.spec_from_file_location('reference_odtk_loss', REF_LOSS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def heads_for(cls_target, box_target, seed, sigma=2.0):
    """Seeded head tensors [B, A*C, H, W] / [B, A*nb, H, W] (float32, CPU)."""
    g = torch.Generator().manual_seed(seed)
    b, a, c, h, w = cls_target.shape
    nb = box_target.shape[2]
    cls = torch.randn(b, a * c, h, w, generator=g) * sigma - 2.0
    box = box_target.reshape(b, a * nb, h, w) + torch.randn(b, a * nb, h, w, generator=g) * 0.15
    return cls, box


def reference_level_loss(ref, cls_head, box_head, cls_target, box_target, depth, dtype):
    """reference odtk/model.py:193-209 for one level, the reference's criteria, evaluated in `dtype`."""
    cls_criterion, box_criterion = ref.FocalLoss(), ref.SmoothL1Loss(beta=0.11)
    cls = cls_head.view_as(cls_target).to(dtype)
    cls_mask = (depth >= 0).expand_as(cls_target).to(dtype)
    cls_loss = (cls_mask * cls_criterion(cls, cls_target.to(dtype))).sum()
    bx = box_head.view_as(box_target).to(dtype)
    box_mask = (depth > 0).expand_as(box_target).to(dtype)
    box_loss = (box_mask * box_criterion(bx, box_target.to(dtype))).sum()
    return cls_loss, box_loss, (depth > 0).sum()


def main():
    ref = reference_loss_module()
    cases = sorted(glob.glob(os.path.join(GOLDEN, 'snap_*.npz'))) + sorted(glob.glob(os.path.join(GOLDEN, 'snaprot_ref_*.npz')))
    for i, path in enumerate(cases):
        with np.load(path) as z:
            cls_target, box_target, depth = (torch.from_numpy(z[k]).unsqueeze(0) for k in ('cls_target', 'box_target', 'depth'))
        cls_head, box_head = heads_for(cls_target, box_target, 100 + i)
        cls_head.requires_grad_(True)
        box_head.requires_grad_(True)
        c32, b32, fg = reference_level_loss(ref, cls_head, box_head, cls_target, box_target, depth, torch.float32)
        (c32 * G_CLS + b32 * G_BOX).backward()
        c64, b64, _ = reference_level_loss(ref, cls_head.detach(), box_head.detach(), cls_target, box_target, depth, torch.float64)
        name = 'loss_ref_' + os.path.basename(path)[:-4].replace('snap_', '').replace('snaprot_ref_', 'rot_')
        np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), targets=os.path.basename(path), cls_head=cls_head.detach().numpy(),
                            box_head=box_head.detach().numpy(), sums32=np.array([float(c32), float(b32), float(fg)], dtype=np.float32),
                            sums64=np.array([float(c64), float(b64), float(fg)], dtype=np.float64), g=np.array([G_CLS, G_BOX]),
                            dcls=cls_head.grad.numpy(), dbox=box_head.grad.numpy())
        print('%-28s cls %.6f box %.6f fg %d' % (name, float(c64), float(b64), int(fg)))


if __name__ == '__main__':
    main()
