"""Import the REFERENCE's own odtk/box.py from /root/reference (build container only).

TEST INFRASTRUCTURE ONLY -- used by oracle/gen_golden.py (to make tests/golden/*.npz)
and by the not-gpu pinning tests when /root/reference exists.  /root/reference does not
exist on the GPU box: nothing on the gpu test / smoke / bench path may call this.

Two shims are needed to run the reference unmodified on torch 2.x:
  1. odtk/box.py:2-4 imports the CUDA extension unconditionally -> pre-register a stub
     `odtk._C` module (and a stub `odtk` package so odtk/__init__.py is not executed).
  2. odtk/box.py:291,296,297 divide int64 index tensors with `/`, relying on torch<1.5
     floor semantics -> temporarily patch Tensor.__truediv__ (integer tensor / python int
     -> floor division) while the reference runs.
"""
import contextlib
import importlib
import os
import sys
import types

import torch

REFERENCE_ROOT = '/root/reference'


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'odtk', 'box.py'))


@contextlib.contextmanager
def legacy_int_division():
    orig = torch.Tensor.__truediv__

    def _legacy(self, other):
        if isinstance(other, int) and not self.is_floating_point() and not self.is_complex():
            return torch.div(self, other, rounding_mode='floor')
        return orig(self, other)

    torch.Tensor.__truediv__ = _legacy
    try:
        yield
    finally:
        torch.Tensor.__truediv__ = orig


@contextlib.contextmanager
def _reference_modules():
    """Temporarily expose the reference as `odtk` in sys.modules, then restore."""
    saved = {k: v for k, v in sys.modules.items() if k == 'odtk' or k.startswith('odtk.')}
    for k in saved:
        del sys.modules[k]
    pkg = types.ModuleType('odtk')
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, 'odtk')]
    stub = types.ModuleType('odtk._C')
    for name in ('decode', 'nms', 'iou', 'Engine'):
        setattr(stub, name, None)
    sys.modules['odtk'] = pkg
    sys.modules['odtk._C'] = stub
    try:
        yield
    finally:
        for k in [k for k in sys.modules if k == 'odtk' or k.startswith('odtk.')]:
            del sys.modules[k]
        sys.modules.update(saved)


_REF_BOX = None


def reference_box():
    """The reference's odtk.box module object (cached; not left in sys.modules)."""
    global _REF_BOX
    if _REF_BOX is None:
        if not available():
            raise RuntimeError('reference tree not present at %s' % REFERENCE_ROOT)
        with _reference_modules():
            _REF_BOX = importlib.import_module('odtk.box')
    return _REF_BOX


def ref_decode(cls_head, box_head, stride, threshold, top_n, anchors):
    """reference odtk/box.py:255-309 (CPU branch), unmodified, under the int-division shim."""
    assert not torch.cuda.is_available(), 'reference would dispatch to the stubbed _C'
    box = reference_box()
    with legacy_int_division():
        return box.decode(cls_head, box_head, stride, threshold, top_n, anchors, False)


def ref_nms(scores, boxes, classes, nms=0.5, ndetections=100):
    """reference odtk/box.py:312-367 (CPU branch), unmodified."""
    assert not torch.cuda.is_available()
    return reference_box().nms(scores, boxes, classes, nms, ndetections)


def ref_generate_anchors(stride, ratios, scales):
    return reference_box().generate_anchors(stride, ratios, scales)


def ref_generate_anchors_rotated(stride, ratios, scales, angles):
    return reference_box().generate_anchors_rotated(stride, ratios, scales, angles)


def ref_snap_to_anchors_rotated(boxes, size, stride, anchors, num_classes, anchor_ious):
    """reference odtk/box.py:192-252, unmodified, on the CPU.  The function only knows a CUDA `iou`
    (box.py:220-223: `if torch.cuda.is_available(): iou = iou_cuda`), so for the duration of the call
    `iou_cuda` is bound to the reference's OWN iou kernel compiled for the CPU (oracle/ref_native.py) and
    `torch.cuda.is_available` answers True; every other line runs as written."""
    from . import ref_native
    box = reference_box()

    def iou_native(boxes_flat, anchors_flat):
        out = ref_native.iou_pairs(boxes_flat.detach().cpu().numpy().reshape(-1, 8),
                                   anchors_flat.detach().cpu().numpy().reshape(-1, 8))
        return [torch.from_numpy(out)]

    saved_iou, saved_avail = box.iou_cuda, torch.cuda.is_available
    box.iou_cuda, torch.cuda.is_available = iou_native, (lambda: True)
    try:
        return box.snap_to_anchors_rotated(boxes, size, stride, anchors, num_classes, 'cpu', anchor_ious)
    finally:
        box.iou_cuda, torch.cuda.is_available = saved_iou, saved_avail
