"""The compiled operator module retinanet-examples_amd/odtk/_C_ext (csrc/odtk_binding.cpp): the reference's pybind
surface (csrc/extensions.cpp:184-201 -- decode, nms, iou, Engine) as a real torch extension over the C ABI, i.e. the
code INTEGRATION.md section 2 asks a maintainer of the reference to write, compiled and exercised:
  * CPU: it imports, exports the reference's names with the reference's positional signatures, and refuses CPU /
    non-contiguous tensors with a RuntimeError like the reference's CHECK_INPUT;
  * GPU: the reference-generated fixture groups (decode, nms, pipeline, rotated nms, iou) pass through it bit for bit,
    and it agrees with the ctypes binding (odtk/_C.py) on fresh inputs."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import box_check
from odtk import _C, box

try:
    from odtk import _C_ext
except ImportError:                      # not built yet (a fresh checkout): build it the way __graft_entry__.build() does
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'retinanet-examples_amd', 'csrc'))
    import build_ext
    build_ext.build()
    from odtk import _C_ext

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]


def _cases(kind):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, '*.npz'))):
        with np.load(p) as z:
            if 'kind' in z.files and str(z['kind']) == kind:
                out.append(p)
    return out


def _load(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def _bits(t, ref, what):
    a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32).view(np.uint32)
    b = np.ascontiguousarray(ref, dtype=np.float32).view(np.uint32)
    assert a.shape == b.shape and np.array_equal(a, b), what


def test_module_is_compiled_against_the_current_header():
    """A stale _C_ext (built before include/odtk_hip.h grew a field) hands the library arrays of the wrong stride: its struct
    sizes must be the ctypes mirror's, which tests/test_abi_host.py ties to the header."""
    import ctypes
    sizes = _C_ext.abi_sizes()
    assert sizes['level'] == ctypes.sizeof(_C.Level)
    assert sizes['snap_level'] == ctypes.sizeof(_C.SnapLevel)
    assert sizes['snap_rot_level'] == ctypes.sizeof(_C.SnapRotLevel)


def test_module_surface_and_errors_on_cpu():
    assert _C_ext.version().startswith('odtk-hip')
    for name in ('decode', 'nms', 'iou', 'detect', 'Engine'):
        assert hasattr(_C_ext, name)
    cls, deltas = torch.rand(1, 36, 3, 3), torch.zeros(1, 36, 3, 3)
    anchors = box.generate_anchors(8, RATIOS, SCALES).view(-1).tolist()
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        _C_ext.decode(cls, deltas, anchors, 8, 0.05, 10)             # reference call shape, box.py:263-264
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        _C_ext.nms(torch.rand(1, 5), torch.rand(1, 5, 4), torch.zeros(1, 5), 0.5, 3, False)
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        _C_ext.iou(torch.rand(8), torch.rand(16))
    with pytest.raises(RuntimeError, match='not available on MI355X'):
        _C_ext.Engine('x', 1)
    with pytest.raises(RuntimeError, match='not available on MI355X'):
        _C_ext.Engine.load('engine.plan')


@pytest.mark.gpu
@pytest.mark.parametrize('path', _cases('decode'), ids=os.path.basename)
def test_ext_decode_vs_reference_fixture(path):
    g = _load(path)
    out = _C_ext.decode(torch.from_numpy(g['cls']).cuda(), torch.from_numpy(g['box']).cuda(), g['anchors'].reshape(-1).tolist(),
                        int(g['stride']), float(g['threshold']), int(g['top_n']))
    _bits(out[0], g['out_scores'], 'scores')
    _bits(out[2], g['out_classes'], 'classes')
    stride = int(g['stride'])
    box_check.check_decode(out[1], g['out_boxes'], [g['cls']], [g['box']], [stride], {stride: torch.from_numpy(g['anchors'])},
                           float(g['threshold']), int(g['top_n']))      # 1e-4, or proven exp rounding (oracle/box_check.py)
    with pytest.raises(RuntimeError, match='must be contiguous'):
        _C_ext.decode(torch.from_numpy(g['cls']).cuda().transpose(2, 3), torch.from_numpy(g['box']).cuda(),
                      g['anchors'].reshape(-1).tolist(), int(g['stride']), float(g['threshold']), int(g['top_n']))


@pytest.mark.gpu
@pytest.mark.parametrize('path', _cases('nms'), ids=os.path.basename)
def test_ext_nms_vs_reference_fixture(path):
    g = _load(path)
    out = _C_ext.nms(torch.from_numpy(g['scores']).cuda(), torch.from_numpy(g['boxes']).cuda(), torch.from_numpy(g['classes']).cuda(),
                     float(g['nms']), int(g['detections']))
    for o, k in zip(out, ('out_scores', 'out_boxes', 'out_classes')):
        _bits(o, g[k], k)


@pytest.mark.gpu
def test_ext_equals_ctypes_binding_incl_rotated_iou_and_detect():
    from odtk import synthetic
    g = torch.Generator().manual_seed(91)
    # rotated decode + nms
    cls, dl, strides = synthetic.pyramid(2, 27, 10, 128, 160, 'dense', 404, num_box=6)
    ang = [-np.pi / 6, 0, np.pi / 6]
    anchors = {s: box.generate_anchors_rotated(s, RATIOS, SCALES, ang) for s in strides}
    for c, d, s in zip(cls, dl, strides):
        a = _C.decode(c.cuda(), d.cuda(), anchors[s][0].reshape(-1).tolist(), s, 0.05, 200, True)
        b = _C_ext.decode(c.cuda(), d.cuda(), anchors[s][0].reshape(-1).tolist(), s, 0.05, 200, True)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        na, nb_ = _C.nms(*a, 0.5, 50, True), _C_ext.nms(*b, 0.5, 50, True)
        for x, y in zip(na, nb_):
            assert torch.equal(x, y)
    # pairwise iou
    quads = torch.rand(7, 8, generator=g).cuda() * 50
    cells = torch.rand(40, 8, generator=g).cuda() * 50
    assert torch.equal(_C.iou(quads.view(-1), cells.view(-1))[0], _C_ext.iou(quads.view(-1), cells.view(-1))[0])
    # batched detect on bf16 channels_last logits
    cls, dl, strides = synthetic.pyramid(2, 9, 20, 128, 192, 'clustered', 4242, unique=False)
    lg = [torch.logit(c.clamp(1e-6, 1 - 1e-6)).cuda().bfloat16().contiguous(memory_format=torch.channels_last) for c in cls]
    db = [d.cuda().bfloat16().contiguous(memory_format=torch.channels_last) for d in dl]
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}
    a = box.detect(lg, db, strides, anchors, 0.05, 1000, 0.5, 100, logits=True)
    b = _C_ext.detect(lg, db, [anchors[s].reshape(-1).tolist() for s in strides], strides, 0.05, 1000, 0.5, 100, False, True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert int((a[0] > 0).sum()) > 50
