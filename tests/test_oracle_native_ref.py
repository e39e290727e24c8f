"""The rotated path pinned to the reference's OWN source: csrc/cuda/nms_iou.cu's device code (Vector /
Line / IntersectionArea, nms_rotate_kernel, iou_cuda_kernel) compiled for the CPU with g++ in IEEE mode
(oracle/ref_build/build_ref.py).  tests/golden/rotated_ref_*.npz hold its outputs; the C restatement
(oracle/c/odtk_oracle.c) must reproduce them bit for bit, and -- when the compiled reference is present
(build container) -- on fresh random inputs as well.  tests/test_gpu_rotated.py then ties the HIP kernels to
the same fixtures.

Convention note: the reference's CUDA path leaves the boxes / classes of SUPPRESSED detections in the output
rows behind the survivors (score 0); this repo follows the CPU path's convention of all-zero padding rows
(SURVEY.md 0.6), so boxes / classes are compared on the kept rows."""
import glob
import os

import numpy as np
import pytest

from oracle import c_oracle, ref_native

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'rotated_ref_iou_*.npz'))), ids=os.path.basename)
def test_c_oracle_pairwise_iou_equals_reference_source(path):
    z = np.load(path)
    got = c_oracle.iou_pairs(z['boxes'], z['anchors'])
    assert got.shape == z['iou'].shape == (z['anchors'].shape[0], z['boxes'].shape[0])
    assert np.array_equal(_bits(got), _bits(z['iou']))
    assert (z['iou'] > 0).mean() > 0.02                  # the fixture is not vacuous


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'rotated_ref_nms_*.npz'))), ids=os.path.basename)
def test_c_oracle_rotated_nms_equals_reference_source(path):
    z = np.load(path)
    ndet = int(z['ndet'])
    s, b, c, idx = c_oracle.nms(z['scores'][None], z['boxes'][None], z['classes'][None], float(z['thresh']), ndet, rotated=True)
    kept = z['out_index'] >= 0
    assert np.array_equal(idx[0], z['out_index']), 'kept positions'
    assert np.array_equal(_bits(s[0]), _bits(z['out_scores']))
    assert np.array_equal(_bits(b[0][kept]), _bits(z['out_boxes'][kept])) and np.array_equal(c[0][kept], z['out_classes'][kept])
    assert not b[0][~kept].any() and not c[0][~kept].any()          # CPU-path padding convention
    assert 0 < kept.sum() and (kept.sum() < (z['scores'] > 0).sum())   # something kept, something suppressed


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'axis_ref_nms_*.npz'))), ids=os.path.basename)
def test_axis_nms_oracles_equal_the_reference_cuda_kernel(path):
    """csrc/cuda/nms.cu:44-80 compiled for the CPU: on tie-free finite inputs the reference's CUDA path and
    its CPU path (normative here) keep the same detections; both oracles reproduce them."""
    import torch
    from oracle import box_oracle
    z = np.load(path)
    ndet, thr = int(z['ndet']), float(z['thresh'])
    kept = z['out_index'] >= 0
    s, b, c, idx = c_oracle.nms(z['scores'][None], z['boxes'][None], z['classes'][None], thr, ndet)
    assert np.array_equal(idx[0], z['out_index']) and np.array_equal(_bits(s[0]), _bits(z['out_scores']))
    assert np.array_equal(_bits(b[0][kept]), _bits(z['out_boxes'][kept]))
    t = box_oracle.nms(torch.from_numpy(z['scores'])[None], torch.from_numpy(z['boxes'])[None],
                       torch.from_numpy(z['classes'])[None], thr, ndet, return_indices=True)
    assert np.array_equal(t[3][0].numpy(), z['out_index']) and np.array_equal(_bits(t[0][0].numpy()), _bits(z['out_scores']))
    assert 0 < kept.sum() < (z['scores'] > 0).sum()


def _check_decode_against_reference_lambda(z, s, b, c, idx):
    """(s, b, c, idx) = one image's decode output of the implementation under test."""
    n = len(z['indices'])
    assert np.array_equal(idx[:n], z['indices']) and (idx[n:] < 0).all()
    assert np.array_equal(_bits(s[:n]), _bits(z['out_scores'])) and np.array_equal(c[:n], z['out_classes'])
    h, w = z['cls'].shape[1:]
    lim = np.array([w, h, w, h], np.float32) * np.float32(z['stride']) - 1
    ref = np.clip(z['out_boxes'][:, :4], 0, lim)                        # the CPU path's two-sided clamp (box.py:105-111)
    assert np.abs(b[:n, :4] - ref).max() <= 1e-4                        # centre-sum order + float exp: a few ulp
    if z['out_boxes'].shape[1] == 6:
        assert np.array_equal(_bits(b[:n, 4:]), _bits(z['out_boxes'][:, 4:]))   # sin, cos pass through untouched
    assert (z['out_boxes'][:, :4] != ref).any()                         # the clamp convention is exercised


@pytest.mark.parametrize('name', ['axis', 'rotated'])
def test_decode_oracles_against_the_reference_cuda_lambda(name):
    """decode.cu:121-159 / decode_rotate.cu:116-167 compiled for the CPU: flat index -> (score, box, class)."""
    z = np.load(os.path.join(GOLDEN, 'decode_ref_%s.npz' % name))
    rotated = name == 'rotated'
    s, b, c, idx = c_oracle.decode(z['cls'][None], z['deltas'][None], int(z['stride']), float(z['thresh']), int(z['top_n']),
                                   z['anchors'], rotated=rotated)
    _check_decode_against_reference_lambda(z, s[0], b[0], c[0], idx[0])
    if not rotated:                                                      # the torch restatement (pinned to box.py) as well
        import torch
        from oracle import box_oracle
        t = box_oracle.decode(torch.from_numpy(z['cls'])[None], torch.from_numpy(z['deltas'])[None], int(z['stride']),
                              float(z['thresh']), int(z['top_n']), torch.from_numpy(z['anchors']), return_indices=True)
        _check_decode_against_reference_lambda(z, t[0][0].numpy(), t[1][0].numpy(), t[2][0].numpy(), t[3][0].numpy())


def test_fixture_set_is_complete():
    assert len(glob.glob(os.path.join(GOLDEN, 'rotated_ref_iou_*.npz'))) == 3
    assert len(glob.glob(os.path.join(GOLDEN, 'rotated_ref_nms_*.npz'))) == 4
    assert len(glob.glob(os.path.join(GOLDEN, 'axis_ref_nms_*.npz'))) == 2
    assert len(glob.glob(os.path.join(GOLDEN, 'decode_ref_*.npz'))) == 2


@pytest.mark.skipif(not ref_native.available(), reason='oracle/_ref/libodtk_ref_native.so not built (needs /root/reference)')
@pytest.mark.parametrize('seed', range(6))
def test_live_against_the_compiled_reference(seed):
    r = np.random.default_rng(900 + seed)
    k = int(r.choice([50, 257, 700]))
    ctr = r.uniform(10, float(r.choice([120, 400])), (k, 2))
    wh = r.uniform(3, 100, (k, 2))
    th = r.uniform(-1.57, 1.57, k)
    amp = r.uniform(0.5, 1.5, k) if seed % 2 else np.ones(k)
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2, (np.sin(th) * amp)[:, None], (np.cos(th) * amp)[:, None]], 1).astype(np.float32)
    scores = ((r.permutation(k) + 1) / k).astype(np.float32)
    scores[r.random(k) < 0.2] = 0
    classes = r.integers(0, int(r.choice([1, 4, 80])), k).astype(np.float32)
    thr, ndet = float(r.choice([0.0, 0.2, 0.5, 0.8])), int(r.choice([20, 100]))
    rs, rb, rc, ri = ref_native.nms_rotate(scores, boxes, classes, thr, ndet)
    s, b, c, idx = c_oracle.nms(scores[None], boxes[None], classes[None], thr, ndet, rotated=True)
    assert np.array_equal(idx[0], ri) and np.array_equal(_bits(s[0]), _bits(rs))
    # pairwise: corners of the same boxes, rotated about their centres
    def corners(bx):
        cx, cy = (bx[:, 0] + bx[:, 2]) / 2, (bx[:, 1] + bx[:, 3]) / 2
        dx = np.stack([bx[:, 0] - cx, bx[:, 2] - cx, bx[:, 2] - cx, bx[:, 0] - cx], 1)
        dy = np.stack([bx[:, 1] - cy, bx[:, 1] - cy, bx[:, 3] - cy, bx[:, 3] - cy], 1)
        x = dx * bx[:, 5:6] - dy * bx[:, 4:5] + cx[:, None]
        y = dy * bx[:, 5:6] + dx * bx[:, 4:5] + cy[:, None]
        return np.stack([x, y], 2).reshape(-1, 8).astype(np.float32)
    q = corners(boxes[:40])
    assert np.array_equal(_bits(c_oracle.iou_pairs(q[:15], q)), _bits(ref_native.iou_pairs(q[:15], q)))
