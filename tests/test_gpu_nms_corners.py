"""Corners the C ABI allows (include/odtk_hip.h limits) that no other GPU test visits -- each against the C restatement
(oracle/c/odtk_oracle.c, pinned to the reference's box.py / nms_iou.cu), kept positions and every output bit for bit:

  * detections_per_im in {301, 1024, 2048} (the kept list outgrows one 1024-candidate round; ODTK_MAX_NMS_DETECTIONS = 2048)
    x {axis-aligned, rotated} x {stand-alone nms on arbitrary input, `detect` (sorted-run mode: nms consumes decode_levels'
    per-level lists as runs), more than ODTK_MAX_NMS_COUNT = 7680 candidates per image (key list in the workspace)};
  * n_levels = 6 = ODTK_MAX_LEVELS in ONE odtk_detect call (other tests use 5, or 7 / 10 split over two calls);
  * rotated decode with top_n = 5000 (> 4096: the 128 KiB dynamic-LDS select_decode variant) and its nms;
  * a `run_len` that does not qualify for the sorted-run mode (top_n < 64) must take the generic path with the same result.
(More than 8 runs cannot be expressed through the ABI: run_len is set by odtk_detect only and ODTK_MAX_LEVELS is 6 --
csrc/nms.hpp carries a static_assert instead of a run-time guard.)
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from odtk import _C, box, synthetic

pytestmark = pytest.mark.gpu

RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]
ANGLES = [-np.pi / 6, 0, np.pi / 6]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_bits(hip, ref, what):
    for k, name in enumerate(('scores', 'boxes', 'classes')):
        assert np.array_equal(bits(hip[k].cpu().numpy()), bits(ref[k])), '%s: %s differ' % (what, name)


def random_candidates(seed, batch, count, n_cls, rotated, spread, quantise=False):
    g = torch.Generator().manual_seed(seed)
    ctr = torch.rand(batch, count, 2, generator=g) * spread + 20
    wh = torch.rand(batch, count, 2, generator=g) * 40 + 2
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 2)
    if rotated:
        th = (torch.rand(batch, count, generator=g) - 0.5) * 3.0
        boxes = torch.cat([boxes, th.sin()[..., None], th.cos()[..., None]], 2)
    scores = torch.rand(batch, count, generator=g)
    if quantise:
        scores = (scores * 64).round() / 64                          # massive ties, some zeros
    classes = torch.randint(0, n_cls, (batch, count), generator=g).float()
    return scores, boxes, classes


@pytest.mark.parametrize('rotated', [False, True], ids=['axis', 'rotated'])
@pytest.mark.parametrize('ndet', [301, 1024, 2048])
@pytest.mark.parametrize('count,n_cls,thr,spread,quantise', [
    (5000, 80, 0.5, 2000.0, False),      # LDS-resident keys, mostly survivors: the kept list passes 1024 and 2048
    (3000, 3, 0.3, 60.0, True),          # heavy suppression + ties: every candidate examined, ~500 kept, many rounds
    (10000, 80, 0.5, 1500.0, False),     # > ODTK_MAX_NMS_COUNT: key list in the workspace
], ids=['lds-5000', 'ties-3000', 'scratch-10000'])
def test_standalone_nms_long_kept_lists(rotated, ndet, count, n_cls, thr, spread, quantise):
    if rotated and count == 10000:
        count = 8000                                                  # (the single-threaded C oracle's polygon clips: keep it in seconds)
    scores, boxes, classes = random_candidates(1000 + ndet + count + rotated, 2, count, n_cls, rotated, spread, quantise)
    out = _C.nms(scores.cuda(), boxes.cuda(), classes.cuda(), thr, ndet, rotated, return_indices=True)
    ref = c_oracle.nms(scores.numpy(), boxes.numpy(), classes.numpy(), thr, ndet, rotated=rotated)
    assert np.array_equal(out[3].cpu().numpy().astype(np.int64), ref[3]), 'kept positions'
    same_bits(out, ref, 'nms')
    kept = int((out[0] > 0).sum(1).max())
    assert kept > 300, kept                                           # the kept list really outgrows the default geometry


def _pyramid(batch, A, C, height, width, kind, seed, nb, strides):
    cls, dl = [], []
    shapes = synthetic.level_shapes(height, width, strides)
    for i, (h, w) in enumerate(shapes):
        lg, d = synthetic.make_level(batch, A, C, h, w, kind, seed + i, nb, stride=strides[i])
        cls.append(lg.sigmoid())
        dl.append(d)
    sizes = [c[0].numel() for c in cls]
    joint = synthetic.make_unique_scores(torch.cat([c.reshape(batch, -1) for c in cls], 1), 0.05)
    cls = [j.reshape(c.shape) for j, c in zip(joint.split(sizes, 1), cls)]
    return cls, dl


def _oracle_detect(cls, dl, strides, anchors, thr, top_n, nms_thr, ndet, rotated):
    per = [c_oracle.decode(c.numpy(), d.numpy(), s, thr, top_n, (anchors[s][0] if rotated else anchors[s]).numpy(), rotated=rotated)
           for c, d, s in zip(cls, dl, strides)]
    cat = [np.concatenate(t, 1) for t in zip(*per)]
    return cat, c_oracle.nms(cat[0], cat[1], cat[2], nms_thr, ndet, rotated=rotated)


@pytest.mark.parametrize('rotated', [False, True], ids=['axis', 'rotated'])
@pytest.mark.parametrize('ndet,top_n', [(301, 1000), (1024, 1000), (2048, 1000), (2048, 2000)],
                         ids=['d301', 'd1024', 'd2048', 'd2048-top2000-scratch-keys'])
def test_detect_sorted_run_mode_long_kept_lists(rotated, ndet, top_n):
    """`detect`: nms reads decode_levels' output as sorted runs.  top_n = 2000 x 5 levels = 10 000 candidates per image puts
    the key list in the workspace (sorted-run mode of the kGlobalKeys kernel)."""
    strides = [8, 16, 32, 64, 128]
    A, nb = (27, 6) if rotated else (9, 4)
    cls, dl = _pyramid(2, A, 12, 256, 384, 'dense', 4100 + ndet + top_n, nb, strides)
    anchors = {s: (box.generate_anchors_rotated(s, RATIOS, SCALES, ANGLES) if rotated else box.generate_anchors(s, RATIOS, SCALES))
               for s in strides}
    out = box.detect([c.cuda() for c in cls], [d.cuda() for d in dl], strides, anchors, 0.05, top_n, 0.5, ndet, rotated)
    cat, ref = _oracle_detect(cls, dl, strides, anchors, 0.05, top_n, 0.5, ndet, rotated)
    same_bits(out, ref, 'detect')
    assert int((out[0] > 0).sum(1).max()) > 300
    # the two-call form on the same candidates (generic mode of nms) agrees with the sorted-run mode
    dec = _C.decode_levels([c.cuda() for c in cls], [d.cuda() for d in dl], [anchors[s][0] if rotated else anchors[s] for s in strides],
                           strides, 0.05, top_n, rotated)
    for k in range(3):
        assert np.array_equal(bits(dec[k].cpu().numpy()), bits(cat[k])), 'decode output %d' % k
    two = _C.nms(dec[0], dec[1], dec[2], 0.5, ndet, rotated)
    for a, b in zip(out, two):
        assert torch.equal(a, b)


@pytest.mark.parametrize('rotated', [False, True], ids=['axis', 'rotated'])
def test_six_levels_in_one_call(rotated):
    """ODTK_MAX_LEVELS = 6 levels through ONE odtk_detect / odtk_decode_levels call (6 sorted runs in nms)."""
    strides = [4, 8, 16, 32, 64, 128]
    A, nb = (27, 6) if rotated else (9, 4)
    cls, dl = _pyramid(3, A, 7, 128, 192, 'dense', 515 + rotated, nb, strides)
    assert len(cls) == 6
    anchors = {s: (box.generate_anchors_rotated(s, RATIOS, SCALES, ANGLES) if rotated else box.generate_anchors(s, RATIOS, SCALES))
               for s in strides}
    out = box.detect([c.cuda() for c in cls], [d.cuda() for d in dl], strides, anchors, 0.05, 300, 0.5, 150, rotated)
    cat, ref = _oracle_detect(cls, dl, strides, anchors, 0.05, 300, 0.5, 150, rotated)
    same_bits(out, ref, 'detect, 6 levels')
    dec = _C.decode_levels([c.cuda() for c in cls], [d.cuda() for d in dl], [anchors[s][0] if rotated else anchors[s] for s in strides],
                           strides, 0.05, 300, rotated, return_indices=True)
    assert np.array_equal(dec[3].cpu().numpy().astype(np.int64), cat[3])
    assert int((out[0] > 0).sum()) > 100


def test_rotated_top_n_5000():
    """Rotated decode with top_n > 4096 (select_decode's 128 KiB dynamic-LDS variant, <6, ..., 16384>) and the nms that
    follows: 2 levels x 5000 = 10 000 candidates per image (workspace key list)."""
    strides = [8, 16]
    cls, dl = _pyramid(2, 27, 10, 192, 256, 'dense', 77, 6, strides)
    anchors = {s: box.generate_anchors_rotated(s, RATIOS, SCALES, ANGLES) for s in strides}
    n_cand = [int((c[0] >= 0.05).sum()) for c in cls]
    assert n_cand[0] > 5000                                            # the cut at 5000 is real on P3
    dec = _C.decode_levels([c.cuda() for c in cls], [d.cuda() for d in dl], [anchors[s][0] for s in strides], strides, 0.05, 5000,
                           True, return_indices=True)
    cat, ref = _oracle_detect(cls, dl, strides, anchors, 0.05, 5000, 0.5, 300, True)
    assert np.array_equal(dec[3].cpu().numpy().astype(np.int64), cat[3]), 'selected indices / order'
    for k in range(3):
        assert np.array_equal(bits(dec[k].cpu().numpy()), bits(cat[k])), 'decode output %d' % k
    out = box.detect([c.cuda() for c in cls], [d.cuda() for d in dl], strides, anchors, 0.05, 5000, 0.5, 300, True)
    same_bits(out, ref, 'detect, rotated, top_n 5000')


@pytest.mark.parametrize('top_n', [37, 63, 64])
def test_short_runs_take_the_generic_path(top_n):
    """run_len < 64 does not qualify for the sorted-run mode (csrc/nms.hpp): same answer through the generic rounds."""
    strides = [8, 16, 32, 64, 128]
    cls, dl = _pyramid(2, 9, 5, 128, 128, 'dense', 900 + top_n, 4, strides)
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}
    out = box.detect([c.cuda() for c in cls], [d.cuda() for d in dl], strides, anchors, 0.05, top_n, 0.5, 100)
    _, ref = _oracle_detect(cls, dl, strides, anchors, 0.05, top_n, 0.5, 100, False)
    same_bits(out, ref, 'detect, top_n %d' % top_n)


@pytest.mark.parametrize('rotated', [False, True], ids=['axis', 'rotated'])
@pytest.mark.parametrize('count,ndet,spread', [(4000, 100, 10.0), (4000, 300, 25.0), (9000, 100, 10.0)],
                         ids=['crowd-4000-d100', 'crowd-4000-d300', 'crowd-9000-scratch-keys'])
def test_heavy_suppression_standalone(rotated, count, ndet, spread):
    """A crowd: thousands of same-class boxes on a few pixels, a handful survive.  Exercises the filter pass (everything
    left is tested against the kept list at once when a round's yield says it will all be examined anyway) and, axis-aligned,
    the push form of a round (one parallel pass per kept box when fewer than 1 in 20 candidates survive)."""
    if rotated and count == 9000:
        count = 8000
    scores, boxes, classes = random_candidates(7000 + count + ndet + rotated, 2, count, 1, rotated, spread)
    g = torch.Generator().manual_seed(count + ndet)
    ctr = (boxes[..., :2] + boxes[..., 2:4]) / 2
    wh = 40 + torch.rand(2, count, 2, generator=g) * 4                # nearly equal sizes: neighbours overlap by more than half
    boxes[..., :2], boxes[..., 2:4] = ctr - wh / 2, ctr + wh / 2
    if rotated:
        th = (torch.rand(2, count, generator=g) - 0.5) * 0.2
        boxes[..., 4], boxes[..., 5] = th.sin(), th.cos()
    out = _C.nms(scores.cuda(), boxes.cuda(), classes.cuda(), 0.5, ndet, rotated, return_indices=True)
    ref = c_oracle.nms(scores.numpy(), boxes.numpy(), classes.numpy(), 0.5, ndet, rotated=rotated)
    assert np.array_equal(out[3].cpu().numpy().astype(np.int64), ref[3]), 'kept positions'
    same_bits(out, ref, 'nms')
    kept = int((out[0] > 0).sum(1).max())
    assert 1 <= kept < count // 20, kept                              # heavy suppression indeed


@pytest.mark.parametrize('rotated', [False, True], ids=['axis', 'rotated'])
def test_heavy_suppression_through_detect(rotated):
    """The same through `detect` (sorted-run mode -> filter -> key list): every level's candidates sit on one small patch."""
    strides = [8, 16, 32]
    A, nb = (27, 6) if rotated else (9, 4)
    g = torch.Generator().manual_seed(31 + rotated)
    cls, dl = [], []
    for s in strides:
        h, w = 256 // s, 320 // s
        c = torch.zeros(2, A * 2, h, w)
        ph, pw = max(2, (3 * h) // 4), max(2, (3 * w) // 4)             # a patch of 3/4 x 3/4 of the level ...
        for a_idx in ((12, 13, 14) if rotated else (3, 4, 5)):          # ... three anchors of one scale (one angle), class 0 only:
            c[:, 2 * a_idx, 1:1 + ph, 1:1 + pw] = torch.rand(2, ph, pw, generator=g) * 0.9 + 0.06   # neighbours overlap by > 1/2
        cls.append(c)
        d = torch.randn(2, A * nb, h, w, generator=g) * 0.05
        if rotated:
            d.view(2, A, nb, h, w)[:, :, 4] = 0.0                    # sin
            d.view(2, A, nb, h, w)[:, :, 5] = 1.0                    # cos: well-formed quads
        dl.append(d)
    sizes = [c[0].numel() for c in cls]
    joint = synthetic.make_unique_scores(torch.cat([c.reshape(2, -1) for c in cls], 1), 0.05)
    cls = [j.reshape(c.shape) for j, c in zip(joint.split(sizes, 1), cls)]
    anchors = {s: (box.generate_anchors_rotated(s, RATIOS, SCALES, ANGLES) if rotated else box.generate_anchors(s, RATIOS, SCALES))
               for s in strides}
    out = box.detect([c.cuda() for c in cls], [d.cuda() for d in dl], strides, anchors, 0.05, 1000, 0.5, 100, rotated)
    cat, ref = _oracle_detect(cls, dl, strides, anchors, 0.05, 1000, 0.5, 100, rotated)
    same_bits(out, ref, 'detect, crowd')
    n_cand = int((cat[0][0] > 0).sum())
    kept = int((out[0][0] > 0).sum())
    assert n_cand > 1500 and 3 <= kept < n_cand // 10, (n_cand, kept)   # (the oracle comparison above is what matters)


@pytest.mark.parametrize('rotated', [False, True], ids=['axis', 'rotated'])
def test_detect_with_a_non_positive_threshold(rotated):
    """score_thresh <= 0: decode emits zero and negative scores too (`score >= thresh`), nms takes `score > 0` only.  In
    `detect` the positive entries of every level's list are a PREFIX whose length select_decode hands over (`run_valid`,
    not the list length): the runs the NMS reads must end there."""
    strides = [8, 16, 32]
    A, nb = (27, 6) if rotated else (9, 4)
    g = torch.Generator().manual_seed(5 + rotated)
    cls, dl = [], []
    for s in strides:
        h, w = 64 // s, 96 // s
        c = torch.randn(2, A * 3, h, w, generator=g) * 0.4                 # signed "scores": about half are <= 0
        c[:, :, ::2, ::2] = 0.0                                             # exact zeros as well
        cls.append(c)
        d = torch.randn(2, A * nb, h, w, generator=g) * 0.3
        dl.append(d)
    sizes = [c[0].numel() for c in cls]
    joint = synthetic.make_unique_scores(torch.cat([c.reshape(2, -1) for c in cls], 1), 1e-6)   # positive scores distinct
    cls = [j.reshape(c.shape) for j, c in zip(joint.split(sizes, 1), cls)]
    anchors = {s: (box.generate_anchors_rotated(s, RATIOS, SCALES, ANGLES) if rotated else box.generate_anchors(s, RATIOS, SCALES))
               for s in strides}
    for thr, top_n in ((-0.25, 400), (0.0, 300), (-10.0, 150)):
        out = box.detect([c.cuda() for c in cls], [d.cuda() for d in dl], strides, anchors, thr, top_n, 0.5, 120, rotated)
        cat, ref = _oracle_detect(cls, dl, strides, anchors, thr, top_n, 0.5, 120, rotated)
        positives = (cat[0] > 0).sum(1)
        assert (cat[0] <= 0).any() and positives.min() > 50            # lists really carry non-positive entries behind the positive ones
        same_bits(out, ref, 'detect, thresh %g' % thr)
        two = _C.nms(*[torch.from_numpy(t).cuda() for t in cat[:3]], 0.5, 120, rotated)
        for a, b in zip(out, two):
            assert torch.equal(a, b)
