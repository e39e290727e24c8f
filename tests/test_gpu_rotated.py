"""Rotated path on the GPU (decode_rotate, rotated NMS, pairwise IoU) against the C restatement of
csrc/cuda/decode_rotate.cu + nms_iou.cu (oracle/c; its rotated IoU / NMS are pinned bit for bit to the
reference's own device code compiled for the CPU, tests/test_oracle_native_ref.py -- the fixtures of
that pinning are also applied to the HIP kernels directly at the end of this file).  Both sides evaluate the same IEEE fp32 expressions in the
same order, so everything -- including IoU values -- is compared BIT FOR BIT; axis-aligned boxes
are also compared bit-for-bit here (the C oracle uses the same correctly rounded exp)."""
import numpy as np
import pytest
import torch

from oracle import box_check, box_oracle, c_oracle
from odtk import _C, box, synthetic
from test_oracle_c import random_box6

pytestmark = pytest.mark.gpu

RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]
ANGLES = [-np.pi / 6, 0, np.pi / 6]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize('kind,seed', [('sparse', 41), ('dense', 42)])
def test_axis_decode_boxes_bit_exact_vs_c_oracle(kind, seed):
    cls, dl, strides = synthetic.pyramid(2, 9, 40, 192, 256, kind, seed)
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}
    out = _C.decode_levels([c.cuda() for c in cls], [d.cuda() for d in dl], [anchors[s] for s in strides], strides,
                           0.05, 500, False, return_indices=True)
    ref = [c_oracle.decode(c.numpy(), d.numpy(), s, 0.05, 500, anchors[s].numpy()) for c, d, s in zip(cls, dl, strides)]
    ref = [np.concatenate(t, 1) for t in zip(*ref)]
    assert np.array_equal(out[3].cpu().numpy().astype(np.int64), ref[3])
    for h, r in zip(out[:3], ref[:3]):
        assert np.array_equal(bits(h.cpu().numpy()), bits(r))


@pytest.mark.parametrize('kind,seed,batch', [('sparse', 51, 2), ('dense', 52, 1)])
def test_rotated_decode_vs_oracles(kind, seed, batch):
    cls, dl, strides = synthetic.pyramid(batch, 27, 10, 128, 192, kind, seed, num_box=6)
    anchors = {s: box.generate_anchors_rotated(s, RATIOS, SCALES, ANGLES) for s in strides}
    per_level = [box.decode(c.cuda(), d.cuda(), s, 0.05, 300, anchors[s], True) for c, d, s in zip(cls, dl, strides)]
    fused = _C.decode_levels([c.cuda() for c in cls], [d.cuda() for d in dl], [anchors[s][0] for s in strides],
                             strides, 0.05, 300, True, return_indices=True)
    cat = [torch.cat(t, 1) for t in zip(*per_level)]
    for a, b in zip(cat, fused[:3]):
        assert torch.equal(a, b)
    assert fused[1].shape[-1] == 6
    ref_c = [c_oracle.decode(c.numpy(), d.numpy(), s, 0.05, 300, anchors[s][0].numpy(), rotated=True)
             for c, d, s in zip(cls, dl, strides)]
    ref_c = [np.concatenate(t, 1) for t in zip(*ref_c)]
    assert np.array_equal(fused[3].cpu().numpy().astype(np.int64), ref_c[3])
    for h, r in zip(fused[:3], ref_c[:3]):
        assert np.array_equal(bits(h.cpu().numpy()), bits(r))
    # and the torch restatement (box.py conventions) within the box tolerance
    ref_t = [box_oracle.decode(c, d, s, 0.05, 300, anchors[s], True) for c, d, s in zip(cls, dl, strides)]
    ref_t = [torch.cat(t, 1) for t in zip(*ref_t)]
    assert torch.equal(fused[0].cpu(), ref_t[0]) and torch.equal(fused[2].cpu(), ref_t[2])
    box_check.check_decode(fused[1], ref_t[1], cls, dl, strides, anchors, 0.05, 300, rotated=True)   # 1e-4, or proven exp rounding
    assert torch.equal(fused[1].cpu()[..., 4:], ref_t[1][..., 4:])     # sin, cos pass through untouched


@pytest.mark.parametrize('count,ndet,thr,own', [(400, 50, 0.3, False), (400, 50, 0.3, True), (3000, 100, 0.5, False),
                                                (5000, 100, 0.5, False), (70, 100, 0.1, False)])
def test_rotated_nms_vs_c_oracle(count, ndet, thr, own):
    rng = np.random.default_rng(count + ndet)
    boxes = np.stack([random_box6(rng, count, spread=200.0) for _ in range(2)])
    scores = np.stack([rng.permutation(count).astype(np.float32) / count + 0.001 for _ in range(2)])
    scores[0, ::5] = 0.0
    classes = rng.integers(0, 4, (2, count)).astype(np.float32)
    lib = _C.library()
    s, b, c = (torch.from_numpy(x).cuda() for x in (scores, boxes, classes))
    if own:
        out = [torch.empty(2, ndet, device='cuda'), torch.empty(2, ndet, 6, device='cuda'),
               torch.empty(2, ndet, device='cuda'), torch.empty(2, ndet, dtype=torch.int32, device='cuda')]
        flags = _C.FLAG_ROTATED | _C.FLAG_ROTATED_NMS_FIXED_ANGLE
        size = lib.odtk_nms_ex(2, None, None, 4, count, ndet, thr, flags, None, 0, None)     # two-phase: query, then call
        assert size > 256                              # rotated: first-round boxes + suppression matrix live in the workspace
        ws = torch.empty(size, dtype=torch.uint8, device='cuda')
        assert lib.odtk_nms_ex(2, _C._ptrs([s, b, c]), _C._ptrs(out), 4, count, ndet, thr, flags, ws.data_ptr(), 256,
                               torch.cuda.current_stream().cuda_stream) == _C.ERR_WORKSPACE   # "Workspace is too small!"
        rc = lib.odtk_nms_ex(2, _C._ptrs([s, b, c]), _C._ptrs(out), 4, count, ndet, thr, flags, ws.data_ptr(), size,
                             torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
    else:
        out = _C.nms(s, b, c, thr, ndet, True, return_indices=True)
        plain = box.nms_rotated(s, b, c, thr, ndet)
        for a, e in zip(plain, out[:3]):
            assert torch.equal(a, e)
    ref = c_oracle.nms(scores, boxes, classes, thr, ndet, rotated=True, own_angle=own)
    assert np.array_equal(out[3].cpu().numpy().astype(np.int64), ref[3])
    for h, r in zip(out[:3], ref[:3]):
        assert np.array_equal(bits(h.cpu().numpy()), bits(r))


def test_pairwise_iou_vs_c_oracle():
    rng = np.random.default_rng(9)
    gt = np.stack([c_oracle.box6_to_quad(b) for b in random_box6(rng, 13, spread=150.0)]).reshape(13, 8).astype(np.float32)
    an = np.stack([c_oracle.box6_to_quad(b) for b in random_box6(rng, 3000, spread=150.0)]).reshape(3000, 8).astype(np.float32)
    out = _C.iou(torch.from_numpy(gt).cuda().view(-1), torch.from_numpy(an).cuda().view(-1))[0]
    assert tuple(out.shape) == (3000, 13)
    ref = c_oracle.iou_pairs(gt, an)
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref))
    # degenerate: identical quads (0.001 pad path) and axis-aligned touching quads
    same = _C.iou(torch.from_numpy(gt).cuda().view(-1), torch.from_numpy(gt).cuda().view(-1))[0]
    assert np.array_equal(bits(same.cpu().numpy()), bits(c_oracle.iou_pairs(gt, gt)))
    # empty inputs are a no-op
    empty = _C.iou(torch.empty(0, device='cuda'), torch.from_numpy(an).cuda().view(-1))[0]
    assert tuple(empty.shape) == (3000, 0)


# ---- fixtures produced by the reference's own device code (oracle/ref_build, tests/test_oracle_native_ref.py) ----
import glob
import os

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(_GOLDEN, 'rotated_ref_iou_*.npz'))), ids=os.path.basename)
def test_hip_pairwise_iou_equals_reference_source_fixture(path):
    z = np.load(path)
    out = _C.iou(torch.from_numpy(z['boxes']).cuda().reshape(-1), torch.from_numpy(z['anchors']).cuda().reshape(-1))[0]
    assert np.array_equal(out.cpu().numpy().view(np.uint32), z['iou'].view(np.uint32))


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(_GOLDEN, 'rotated_ref_nms_*.npz'))), ids=os.path.basename)
def test_hip_rotated_nms_equals_reference_source_fixture(path):
    z = np.load(path)
    out = _C.nms(torch.from_numpy(z['scores'])[None].cuda(), torch.from_numpy(z['boxes'])[None].cuda(),
                 torch.from_numpy(z['classes'])[None].cuda(), float(z['thresh']), int(z['ndet']), True, return_indices=True)
    kept = z['out_index'] >= 0
    assert np.array_equal(out[3][0].cpu().numpy().astype(np.int64), z['out_index'])
    assert np.array_equal(out[0][0].cpu().numpy().view(np.uint32), z['out_scores'].view(np.uint32))
    assert np.array_equal(out[1][0].cpu().numpy()[kept].view(np.uint32), z['out_boxes'][kept].view(np.uint32))
    assert np.array_equal(out[2][0].cpu().numpy()[kept], z['out_classes'][kept])


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(_GOLDEN, 'axis_ref_nms_*.npz'))), ids=os.path.basename)
def test_hip_axis_nms_equals_reference_cuda_kernel_fixture(path):
    z = np.load(path)
    out = _C.nms(torch.from_numpy(z['scores'])[None].cuda(), torch.from_numpy(z['boxes'])[None].cuda(),
                 torch.from_numpy(z['classes'])[None].cuda(), float(z['thresh']), int(z['ndet']), False, return_indices=True)
    kept = z['out_index'] >= 0
    assert np.array_equal(out[3][0].cpu().numpy().astype(np.int64), z['out_index'])
    assert np.array_equal(out[0][0].cpu().numpy().view(np.uint32), z['out_scores'].view(np.uint32))
    assert np.array_equal(out[1][0].cpu().numpy()[kept].view(np.uint32), z['out_boxes'][kept].view(np.uint32))


@pytest.mark.parametrize('name', ['axis', 'rotated'])
def test_hip_decode_against_the_reference_cuda_lambda(name):
    from test_oracle_native_ref import _check_decode_against_reference_lambda
    z = np.load(os.path.join(_GOLDEN, 'decode_ref_%s.npz' % name))
    rotated = name == 'rotated'
    out = _C.decode_levels([torch.from_numpy(z['cls'])[None].cuda()], [torch.from_numpy(z['deltas'])[None].cuda()],
                           [torch.from_numpy(z['anchors'])], [int(z['stride'])], float(z['thresh']), int(z['top_n']), rotated,
                           return_indices=True)
    _check_decode_against_reference_lambda(z, out[0][0].cpu().numpy(), out[1][0].cpu().numpy(), out[2][0].cpu().numpy(),
                                           out[3][0].cpu().numpy().astype(np.int64))
