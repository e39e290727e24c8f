"""Training-side target assignment: the oracle restatement against the
reference's own snap_to_anchors (golden fixtures + live), and the product implementation against
the oracle."""
import glob
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import box_oracle, ref_loader
from odtk import box

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
SNAP = sorted(p for p in glob.glob(os.path.join(GOLDEN, 'snap_*.npz')))
RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]


def _bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32)


@pytest.mark.parametrize('path', SNAP, ids=os.path.basename)
def test_snap_to_anchors_matches_reference_fixture(path):
    with np.load(path) as z:
        g = {k: z[k] for k in z.files}
    boxes, anchors = torch.from_numpy(g['boxes']).view(-1, 5), torch.from_numpy(g['anchors'])
    size, stride, classes, ious = [int(v) for v in g['size']], int(g['stride']), int(g['classes']), list(g['ious'])
    ora = box_oracle.snap_to_anchors(boxes, size, stride, anchors, classes, ious)
    prod = box.snap_to_anchors(boxes, size, stride, anchors, classes, 'cpu', ious)
    for o, p, k in zip(ora, prod, ('cls_target', 'box_target', 'depth')):
        assert o.shape == tuple(g[k].shape)
        assert np.array_equal(_bits(o.numpy()), _bits(p.numpy())), k     # oracle == product on this machine, bit for bit
        if k == 'box_target':
            # the deltas go through torch.log, whose vectorised CPU routine differs by an ulp between CPU generations (the
            # fixture was made in the build container; the GPU box's host is another CPU): values within 2 ulp, the decisions
            # (which anchors got a box, i.e. exact zeros) bit for bit
            assert np.array_equal(o.numpy() == 0, g[k] == 0), k
            assert np.allclose(o.numpy(), g[k], rtol=3e-7, atol=1e-7), k
        else:
            assert np.array_equal(_bits(o.numpy()), _bits(g[k])), k      # == reference, bit for bit


@pytest.mark.skipif(not ref_loader.available() or torch.cuda.is_available(), reason='reference tree only in the build container')
def test_snap_to_anchors_live_reference():
    warnings.filterwarnings('ignore')
    g = torch.Generator().manual_seed(9)
    for stride, size, n in [(8, (320, 256), 30), (32, (320, 256), 5), (128, (384, 256), 2)]:
        anchors = box.generate_anchors(stride, RATIOS, SCALES)
        xy = torch.rand(n, 2, generator=g) * torch.tensor([size[0] * 0.7, size[1] * 0.7])
        wh = torch.rand(n, 2, generator=g) * torch.tensor([size[0] * 0.6, size[1] * 0.6]) + 4
        boxes = torch.cat([xy, wh, torch.randint(0, 80, (n, 1), generator=g).float()], 1)
        ref = ref_loader.reference_box().snap_to_anchors(boxes, list(size), stride, anchors, 80, 'cpu', [0.4, 0.5])
        got = box.snap_to_anchors(boxes, list(size), stride, anchors, 80, 'cpu', [0.4, 0.5])
        for r, o in zip(ref, got):
            assert np.array_equal(_bits(r.numpy()), _bits(o.numpy()))


def test_rotate_boxes_geometry():
    b = torch.tensor([[10., 20., 40., 10., 0.0], [10., 20., 40., 10., np.pi / 2], [0., 0., 8., 8., 0.3]])
    axis, quads = box.rotate_boxes(b)
    assert torch.allclose(axis[0], torch.tensor([10., 20., 49., 29., 0., 1.]))
    q0 = quads[0].view(4, 2)
    assert torch.allclose(q0, torch.tensor([[10., 20.], [50., 20.], [50., 30.], [10., 30.]]))        # tl, tr, br, bl
    # rotation preserves side lengths and the centre
    for i in range(3):
        q = quads[i].view(4, 2)
        assert torch.allclose(q.mean(0), torch.tensor([b[i, 0] + b[i, 2] / 2, b[i, 1] + b[i, 3] / 2]), atol=1e-4)
        sides = (q - q.roll(-1, 0)).norm(dim=1)
        assert torch.allclose(sides.sort().values, torch.tensor([b[i, 2:4].min()] * 2 + [b[i, 2:4].max()] * 2), atol=1e-4)


def test_loss_pipeline_on_cpu():
    """Model._compute_loss end to end on CPU (pure torch part of the training path)."""
    from odtk.model import Model
    torch.manual_seed(0)
    m = Model('ResNet18FPN', classes=5)
    m.initialize(None)
    m.train()
    x = torch.randn(2, 3, 128, 128)
    targets = torch.tensor([[[10., 10., 60., 50., 2.], [70., 30., 40., 80., 4.], [-1, -1, -1, -1, -1]],
                            [[5., 60., 100., 40., 0.], [-1, -1, -1, -1, -1], [-1, -1, -1, -1, -1]]])
    cls_loss, box_loss = m([x, targets])
    assert torch.isfinite(cls_loss) and torch.isfinite(box_loss) and cls_loss > 0
    (cls_loss + box_loss).backward()
    g = m.cls_head[-1].weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0


# ------------------------------------------------------------------------------------------------
# fused HIP kernel (csrc/targets.hpp) -- GPU
# ------------------------------------------------------------------------------------------------
def _pad(boxes_list, n_max):
    out = torch.full((len(boxes_list), n_max, 5), -1.0)
    for i, b in enumerate(boxes_list):
        out[i, :b.shape[0]] = b
    return out


@pytest.mark.gpu
@pytest.mark.parametrize('path', SNAP, ids=os.path.basename)
def test_fused_kernel_matches_reference_fixture(path):
    from odtk import _C
    with np.load(path) as z:
        g = {k: z[k] for k in z.files}
    boxes, anchors = torch.from_numpy(g['boxes']).view(-1, 5), torch.from_numpy(g['anchors'])
    size, stride, classes, ious = [int(v) for v in g['size']], int(g['stride']), int(g['classes']), list(g['ious'])
    w, h = size[0] // stride, size[1] // stride
    targets = _pad([boxes, boxes[:0]], max(boxes.shape[0], 1) + 3).cuda()          # image 1 has no boxes
    cls, box_t, depth = _C.snap_to_anchors(targets, anchors, classes, h, w, stride, ious[0], ious[1])
    assert np.array_equal(_bits(cls[0].cpu().numpy()), _bits(g['cls_target']))
    assert np.array_equal(_bits(depth[0].cpu().numpy()), _bits(g['depth']))
    # deltas: only log() differs from the CPU reference (<= 1-2 ulp)
    assert np.allclose(box_t[0].cpu().numpy(), g['box_target'], rtol=1e-6, atol=1e-6)
    assert float(cls[1].abs().sum()) == 0 and float(box_t[1].abs().sum()) == 0 and float(depth[1].abs().sum()) == 0


@pytest.mark.gpu
def test_fused_kernel_random_batches_vs_oracle():
    from odtk import _C
    g = torch.Generator().manual_seed(77)
    for stride, (wpx, hpx), n_list in [(8, (320, 256), [30, 1, 0, 17]), (32, (352, 224), [5, 12]), (128, (384, 256), [2])]:
        anchors = box.generate_anchors(stride, RATIOS, SCALES)
        per_image = []
        for n in n_list:
            xy = torch.rand(n, 2, generator=g) * torch.tensor([wpx * 0.7, hpx * 0.7])
            wh = torch.rand(n, 2, generator=g) * torch.tensor([wpx * 0.6, hpx * 0.6]) + 4
            per_image.append(torch.cat([xy, wh, torch.randint(0, 80, (n, 1), generator=g).float()], 1))
        # padding rows interleaved (not only at the end): the kernel filters by class > -1 in order
        targets = _pad(per_image, max(n_list) + 2)
        targets[0, [1, 3]] = targets[0, [3, 1]]
        per_image[0] = targets[0][targets[0][:, 4] > -1]
        out = _C.snap_to_anchors(targets.cuda(), anchors, 80, hpx // stride, wpx // stride, stride, 0.4, 0.5)
        for i, b in enumerate(per_image):
            ora = box_oracle.snap_to_anchors(b, [wpx, hpx], stride, anchors, 80, [0.4, 0.5])
            assert np.array_equal(_bits(out[0][i].cpu().numpy()), _bits(ora[0].numpy())), (stride, i, 'cls')
            assert np.array_equal(_bits(out[2][i].cpu().numpy()), _bits(ora[2].numpy())), (stride, i, 'depth')
            assert np.allclose(out[1][i].cpu().numpy(), ora[1].numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_training_loss_fused_equals_torch_targets():
    """Model._compute_loss with the fused HIP target assignment == with the per-image torch path (rotated=False
    models take the kernel for ANY number of target rows; the torch path is forced by patching the dispatcher)."""
    from odtk.model import Model
    from odtk import train as T
    torch.manual_seed(0)
    m = Model('ResNet18FPN', classes=6)
    m.initialize(None)
    m = m.cuda().train()
    m.fused_loss = False                                   # this test is about the targets: torch losses on both sides
    data, target = T.SyntheticBatches(2, 256, 320, classes=6, max_boxes=6, seed=5, device='cuda').batch()
    with torch.no_grad():
        heads = m.heads(data)
        fused = m._compute_loss(data, *heads, target)
        real = box.snap_to_anchors_batched
        try:
            box.snap_to_anchors_batched = lambda t, w, h, s, a, c, i, want=True: tuple(
                torch.stack(p) for p in zip(*[box.snap_to_anchors(r[r[:, -1] > -1], [w * s, h * s], s, a.to(t.device), c, t.device, i)
                                              for r in t]))
            plain = m._compute_loss(data, *heads, target)
        finally:
            box.snap_to_anchors_batched = real
    assert torch.allclose(fused[0], plain[0], rtol=1e-6) and torch.allclose(fused[1], plain[1], rtol=1e-5)


@pytest.mark.gpu
def test_fused_kernel_more_than_1024_rows():
    """n_max > 1024: the rows go through LDS in rounds; first maximum over ALL rows wins, as torch.max."""
    from odtk import _C
    g = torch.Generator().manual_seed(3)
    stride, wpx, hpx, n = 16, 320, 256, 2500
    anchors = box.generate_anchors(stride, RATIOS, SCALES)
    xy = torch.rand(n, 2, generator=g) * torch.tensor([wpx * 0.8, hpx * 0.8])
    wh = torch.rand(n, 2, generator=g) * 80 + 8
    boxes = torch.cat([xy, wh, torch.randint(0, 80, (n, 1), generator=g).float()], 1)
    boxes[1500] = boxes[20]                                 # an exact duplicate in a later round: the FIRST must win
    boxes[1500, 4] = (boxes[20, 4] + 1) % 80
    targets = _pad([boxes, boxes[:1100]], n + 5)
    out = _C.snap_to_anchors(targets.cuda(), anchors, 80, hpx // stride, wpx // stride, stride, 0.4, 0.5)
    nocls = _C.snap_to_anchors(targets.cuda(), anchors, 80, hpx // stride, wpx // stride, stride, 0.4, 0.5, want_cls_target=False)
    assert nocls[0] is None and torch.equal(nocls[1], out[1]) and torch.equal(nocls[2], out[2])
    for i, b in enumerate([boxes, boxes[:1100]]):
        ora = box_oracle.snap_to_anchors(b, [wpx, hpx], stride, anchors, 80, [0.4, 0.5])
        assert np.array_equal(_bits(out[0][i].cpu().numpy()), _bits(ora[0].numpy()))
        assert np.array_equal(_bits(out[2][i].cpu().numpy()), _bits(ora[2].numpy()))
        assert np.allclose(out[1][i].cpu().numpy(), ora[1].numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('want_cls', [True, False])
def test_all_levels_in_one_launch_equal_the_per_level_launches(want_cls):
    """odtk_snap_to_anchors_levels (one launch over a level table, what the fused training loss uses) writes the same bits
    as five odtk_snap_to_anchors launches -- which are pinned to the reference's fixtures above; 6 levels = ODTK_MAX_LEVELS,
    one of them a single cell, padding rows, an image without any box."""
    from odtk import _C
    g = torch.Generator().manual_seed(8)
    strides, wpx, hpx = [4, 8, 16, 32, 64, 128], 320, 256
    sizes = [(max(1, hpx // s), max(1, wpx // s)) for s in strides]
    anchors = [box.generate_anchors(s, RATIOS, SCALES) for s in strides]
    per_image = []
    for n in (7, 0, 31):
        xy = torch.rand(n, 2, generator=g) * torch.tensor([wpx * 0.8, hpx * 0.8])
        wh = torch.rand(n, 2, generator=g) * 120 + 6
        per_image.append(torch.cat([xy, wh, torch.randint(0, 12, (n, 1), generator=g).float()], 1))
    targets = _pad(per_image, 40).cuda()
    cls_all, box_all, depth_all = _C.snap_to_anchors_levels(targets, anchors, 12, sizes, strides, 0.4, 0.5, want_cls_target=want_cls)
    for l, (s, (h, w)) in enumerate(zip(strides, sizes)):
        one = _C.snap_to_anchors(targets, anchors[l], 12, h, w, s, 0.4, 0.5, want_cls_target=want_cls)
        assert (cls_all[l] is None) == (not want_cls)
        if want_cls:
            assert torch.equal(cls_all[l], one[0]), 'level %d class map' % l
        assert torch.equal(box_all[l].view(torch.int32), one[1].view(torch.int32)), 'level %d deltas' % l
        assert torch.equal(depth_all[l], one[2]), 'level %d depth' % l
    assert float(depth_all[1][0].max()) > 0 and float(depth_all[1][1].max()) == 0       # image 1 has no box: all background


# ---- rotated target assignment against the reference's OWN snap_to_anchors_rotated (odtk/box.py:192-252), run on the
# CPU with its iou_cuda bound to its own iou kernel compiled for the CPU (oracle/ref_loader.py, oracle/ref_native.py) ----
SNAPROT = sorted(glob.glob(os.path.join(GOLDEN, 'snaprot_ref_*.npz')))
ANGLES = [-np.pi / 6, 0, np.pi / 6]


def _oracle_iou(boxes_flat, anchors_flat):
    from oracle import c_oracle           # bit-identical to the HIP op (tests/test_gpu_rotated.py) and to the reference kernel
    return [torch.from_numpy(c_oracle.iou_pairs(boxes_flat.cpu().numpy().reshape(-1, 8), anchors_flat.cpu().numpy().reshape(-1, 8)))]


def _rotated_case(path):
    with np.load(path) as z:
        g = {k: z[k] for k in z.files}
    stride = int(g['stride'])
    anchors = box.generate_anchors_rotated(stride, RATIOS, SCALES, ANGLES)
    return g, torch.from_numpy(g['boxes']).view(-1, 6), [int(v) for v in g['size']], stride, anchors, int(g['classes']), [float(v) for v in g['ious']]


@pytest.mark.parametrize('path', SNAPROT, ids=os.path.basename)
def test_snap_to_anchors_rotated_matches_reference_fixture(path, monkeypatch):
    """The torch logic of the product function (rotate_boxes, cell anchors, arg-max, deltas, depth, class map) with
    the HIP iou op stood in by the oracle: equal to the reference's function bit for bit."""
    assert len(SNAPROT) == 3
    g, boxes, size, stride, anchors, classes, ious = _rotated_case(path)
    monkeypatch.setattr(box, '_require_gpu', lambda *a: None)
    monkeypatch.setattr(box._C, 'iou', _oracle_iou)
    out = box.snap_to_anchors_rotated(boxes, size, stride, anchors, classes, 'cpu', ious)
    for o, k in zip(out, ('cls_target', 'box_target', 'depth')):
        assert o.shape == tuple(g[k].shape) and np.array_equal(_bits(o.numpy()), _bits(g[k])), k


@pytest.mark.skipif(not ref_loader.available() or torch.cuda.is_available(), reason='reference tree only in the build container')
def test_snap_to_anchors_rotated_live_reference(monkeypatch):
    from oracle import ref_native
    if not ref_native.available():
        pytest.skip('oracle/_ref not built')
    warnings.filterwarnings('ignore')
    monkeypatch.setattr(box, '_require_gpu', lambda *a: None)
    monkeypatch.setattr(box._C, 'iou', _oracle_iou)
    g = torch.Generator().manual_seed(17)
    for stride, size, n in [(8, (96, 64), 4), (64, (320, 256), 9)]:
        xy = torch.rand(n, 2, generator=g) * torch.tensor([size[0] * 0.7, size[1] * 0.7])
        wh = torch.rand(n, 2, generator=g) * 50 + 10
        boxes = torch.cat([xy, wh, (torch.rand(n, 1, generator=g) - 0.5) * 1.5, torch.randint(0, 20, (n, 1), generator=g).float()], 1)
        ref = ref_loader.ref_snap_to_anchors_rotated(boxes, list(size), stride,
                                                     ref_loader.ref_generate_anchors_rotated(stride, RATIOS, SCALES, ANGLES), 20, [0.4, 0.5])
        got = box.snap_to_anchors_rotated(boxes, list(size), stride, box.generate_anchors_rotated(stride, RATIOS, SCALES, ANGLES),
                                          20, 'cpu', [0.4, 0.5])
        for r, o in zip(ref, got):
            assert np.array_equal(_bits(r.numpy()), _bits(o.numpy()))


@pytest.mark.gpu
@pytest.mark.parametrize('path', SNAPROT, ids=os.path.basename)
def test_snap_to_anchors_rotated_on_gpu_matches_reference_fixture(path):
    """The real path: HIP iou op + torch on the GPU.  Class map and depth exact; deltas differ from the CPU
    reference only through the GPU's log()."""
    g, boxes, size, stride, anchors, classes, ious = _rotated_case(path)
    out = box.snap_to_anchors_rotated(boxes.cuda(), size, stride, anchors, classes, 'cuda', ious)
    assert np.array_equal(_bits(out[0].cpu().numpy()), _bits(g['cls_target']))
    assert np.array_equal(_bits(out[2].cpu().numpy()), _bits(g['depth']))
    assert np.allclose(out[1].cpu().numpy(), g['box_target'], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('path', SNAPROT, ids=os.path.basename)
def test_fused_rotated_assignment_matches_reference_fixture(path):
    """SURVEY 8(f2), rotated: ONE HIP launch for all levels and images (csrc/targets.hpp: polygon IoU per anchor cell with a
    distance reject in front of the clip, arg-max, deltas, depth, class map -- no [27*H*W, N] IoU matrix) against the
    reference's own snap_to_anchors_rotated output: class map and depth bit for bit, deltas up to the GPU's log().  Image 1
    of the batch has no boxes (all padding rows): zeros, as the reference returns for an empty target."""
    g, boxes, size, stride, anchors, classes, ious = _rotated_case(path)
    n = boxes.shape[0]
    targets = torch.full((2, n + 3, 6), -1.0)
    targets[0, :n] = boxes
    if n:                                                       # padding rows in between are skipped, order is kept
        targets[0, n + 1] = boxes[n - 1]
        targets[0, n - 1] = -1.0
    h, w = int(size[1] / stride), int(size[0] / stride)
    cls_t, box_t, depth = box.snap_to_anchors_rotated_levels(targets.cuda(), [(h, w)], [stride], [anchors], classes, ious)
    assert np.array_equal(_bits(cls_t[0][0].cpu().numpy()), _bits(g['cls_target']))
    assert np.array_equal(_bits(depth[0][0].cpu().numpy()), _bits(g['depth']))
    assert np.allclose(box_t[0][0].cpu().numpy(), g['box_target'], rtol=1e-5, atol=1e-5)
    for t in (cls_t[0][1], box_t[0][1], depth[0][1]):
        assert not t.any()
    # without the class map (what the fused loss asks for)
    none, box2, depth2 = box.snap_to_anchors_rotated_levels(targets.cuda(), [(h, w)], [stride], [anchors], classes, ious, want_cls_target=False)
    assert none[0] is None and torch.equal(box2[0], box_t[0]) and torch.equal(depth2[0], depth[0])


@pytest.mark.gpu
def test_fused_rotated_assignment_equals_the_per_image_path_on_every_level():
    """All levels of a 256 x 320 image, two images with different box counts, boxes large and small, at the borders and far
    outside: the fused launch equals the per-image path (HIP iou op over ALL pairs + torch), i.e. the distance reject never
    changes a value."""
    gen = torch.Generator().manual_seed(23)
    strides = [8, 16, 32, 64, 128]
    size = (320, 256)
    anchors = [box.generate_anchors_rotated(s, RATIOS, SCALES, ANGLES) for s in strides]
    sizes = [(-(-size[1] // s), -(-size[0] // s)) for s in strides]
    n = 11
    xy = torch.rand(2, n, 2, generator=gen) * torch.tensor([size[0] * 1.2, size[1] * 1.2]) - 30
    wh = torch.cat([torch.rand(2, n, 1, generator=gen) * 200 + 2, torch.rand(2, n, 1, generator=gen) * 120 + 2], 2)
    theta = (torch.rand(2, n, 1, generator=gen) - 0.5) * 3.0
    cls = torch.randint(0, 20, (2, n, 1), generator=gen).float()
    targets = torch.cat([xy, wh, theta, cls], 2)
    targets[1, 6:] = -1.0
    cls_t, box_t, depth = box.snap_to_anchors_rotated_levels(targets.cuda(), sizes, strides, anchors, 20, [0.4, 0.5])
    hits = 0
    for lvl, (s, (h, w)) in enumerate(zip(strides, sizes)):
        for b in range(2):
            t = targets[b][targets[b][:, -1] > -1]
            ref = box.snap_to_anchors_rotated(t.cuda(), [w * s, h * s], s, anchors[lvl], 20, 'cuda', [0.4, 0.5])
            assert torch.equal(cls_t[lvl][b], ref[0]) and torch.equal(depth[lvl][b], ref[2]), (lvl, b)
            assert torch.allclose(box_t[lvl][b], ref[1], rtol=1e-5, atol=1e-5), (lvl, b)
            hits += int((ref[2] > 0).sum())
    assert hits > 20
