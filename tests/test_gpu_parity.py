"""Parity tests proper (need an MI355X): the HIP path, called through the C ABI (ctypes ->
libodtk_hip.so), against
  (1) the committed golden fixtures = outputs of the REFERENCE's own odtk/box.py, and
  (2) the pinned oracle (oracle/box_oracle.py) on fresh seeded inputs and edge cases.

The bar (BASELINE.json north_star): selection/order (indices) and classes BIT-EXACT, scores
BIT-EXACT (they are passed through), box coordinates within 1e-4.  The only non-bit-exact operation
is exp() (torch-CPU's vectorised expf vs a correctly rounded expf on the GPU, at most 1 ulp apart):
where an ulp of the coordinate exceeds 1e-4 (|coordinate| >= 512 px) a deviation beyond the bar is
accepted ONLY with proof that it is the reference's own exp rounding (oracle/box_check.py: bit-equal
to the C restatement and no further from the float64 evaluation of box.py:97-111 than the reference).
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import box_check, box_oracle
from odtk import _C, box, synthetic

pytestmark = pytest.mark.gpu

RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


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


def _np(t):
    return t.detach().float().cpu().numpy()


def assert_bits(hip, ref, what):
    h, r = np.ascontiguousarray(_np(hip) if isinstance(hip, torch.Tensor) else hip, dtype=np.float32), \
        np.ascontiguousarray(_np(ref) if isinstance(ref, torch.Tensor) else ref, dtype=np.float32)
    assert h.shape == r.shape, what
    bad = np.flatnonzero(h.view(np.uint32).ravel() != r.view(np.uint32).ravel())
    assert bad.size == 0, '%s: %d/%d elements differ (first at %d: %r vs %r)' % (
        what, bad.size, h.size, bad[0], h.ravel()[bad[0]], r.ravel()[bad[0]])


def assert_boxes(hip, ref, what, proof=None):
    """1e-4, the north star's bar.  `proof` = (exact, truth) providers (oracle/box_check.py): a coordinate beyond the bar is
    accepted only when it is bit-equal to the C restatement and no further from the float64 evaluation of box.py:97-111 than
    the reference's own value -- i.e. shown to be the reference's exp rounding."""
    h = _np(hip) if isinstance(hip, torch.Tensor) else np.asarray(hip, np.float32)
    r = _np(ref) if isinstance(ref, torch.Tensor) else np.asarray(ref, np.float32)
    assert h.shape == r.shape, what
    assert np.all(np.isfinite(h) == np.isfinite(r)), what
    finite = np.isfinite(r)
    exact, truth = proof if proof is not None else (None, None)
    return box_check.check_boxes(np.where(finite, h, 0), np.where(finite, r, 0), exact, truth, what)


def decode_proof(g):
    """Lazy proof inputs for a one-level decode fixture made by the reference."""
    stride = int(g['stride'])
    return [box_check.ImageProof([g['cls'][b:b + 1]], [g['box'][b:b + 1]], [stride], {stride: torch.from_numpy(g['anchors'])},
                                 float(g['threshold']), int(g['top_n'])) for b in range(g['cls'].shape[0])]


def cuda(t):
    return torch.as_tensor(t).cuda()


# ------------------------------------------------------------------------------------------------
# (1) golden fixtures produced by the reference itself
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('path', _cases('decode'), ids=os.path.basename)
def test_decode_vs_reference_fixture(path):
    g = _load(path)
    out = box.decode(cuda(g['cls']), cuda(g['box']), int(g['stride']), float(g['threshold']), int(g['top_n']),
                     torch.from_numpy(g['anchors']))
    assert_bits(out[0], g['out_scores'], 'scores')
    assert_bits(out[2], g['out_classes'], 'classes')
    for b, proof in enumerate(decode_proof(g)):
        assert_boxes(out[1][b], g['out_boxes'][b], 'boxes of image %d' % b, (proof.exact, proof.truth))


@pytest.mark.parametrize('path', _cases('nms'), ids=os.path.basename)
def test_nms_vs_reference_fixture(path):
    g = _load(path)
    out = box.nms(cuda(g['scores']), cuda(g['boxes']), cuda(g['classes']), float(g['nms']), int(g['detections']))
    assert_bits(out[0], g['out_scores'], 'scores')
    assert_bits(out[1], g['out_boxes'], 'boxes')       # nms copies boxes: bit-exact
    assert_bits(out[2], g['out_classes'], 'classes')


@pytest.mark.parametrize('path', _cases('nms_ties'), ids=os.path.basename)
def test_nms_on_a_trained_detectors_tied_candidates(path):
    """The NMS input of a TRAINED detector's bf16 engine (clusters of overlapping same-class candidates, 16-bit scores tying in the
    hundreds; oracle/gen_golden_trained_nms.py): the canonical rule (score desc, position asc), bit for bit -- through the
    stand-alone op (arbitrary order in, generic rounds) and with the candidates handed over in `detect`'s sorted-run form."""
    from odtk import _C
    g = _load(path)
    args = (cuda(g['scores']), cuda(g['boxes']), cuda(g['classes']))
    run_len = g['scores'].shape[1] // 5                      # five levels of top_n candidates, as decode_levels wrote them
    for name, out in (('nms', box.nms(*args, float(g['nms']), int(g['detections']))),
                      ('nms_sorted_runs', _C.nms_sorted_runs(*args, run_len, float(g['nms']), int(g['detections'])))):
        assert_bits(out[0], g['out_scores'], name + ' scores')
        assert_bits(out[1], g['out_boxes'], name + ' boxes')
        assert_bits(out[2], g['out_classes'], name + ' classes')


def test_nms_sorted_runs_equals_nms_on_decode_output():
    """odtk_nms_sorted_runs on what decode_levels wrote == odtk_nms_ex on the same candidates == detect (which calls the same code)."""
    from odtk import _C
    cls, dl, strides = synthetic.pyramid(3, 9, 20, 256, 320, 'clustered', 77)
    anchors = {s: box.generate_anchors(s, [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)]) for s in strides}
    dec = box.decode_levels([cuda(c) for c in cls], [cuda(d) for d in dl], strides, 0.05, 300, anchors)
    a = box.nms(dec[0], dec[1], dec[2], 0.5, 100)
    b = _C.nms_sorted_runs(dec[0], dec[1], dec[2], 300, 0.5, 100)
    c = box.detect([cuda(c) for c in cls], [cuda(d) for d in dl], strides, anchors, 0.05, 300, 0.5, 100)
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)


@pytest.mark.parametrize('path', _cases('pipeline'), ids=os.path.basename)
def test_pipeline_vs_reference_fixture(path):
    g = _load(path)
    strides = [int(s) for s in g['strides']]
    n = len(strides)
    cls = [cuda(g['cls%d' % i]) for i in range(n)]
    dl = [cuda(g['box%d' % i]) for i in range(n)]
    anchors = {s: torch.from_numpy(g['anchors%d' % i]) for i, s in enumerate(strides)}
    thr, top_n, nms, det = float(g['threshold']), int(g['top_n']), float(g['nms']), int(g['detections'])
    # (a) per-level calls + torch.cat, exactly as reference model.py:153-165 does
    per_level = [box.decode(c, d, s, thr, top_n, anchors[s]) for c, d, s in zip(cls, dl, strides)]
    cat = [torch.cat(t, 1) for t in zip(*per_level)]
    assert_bits(cat[0], g['cat_scores'], 'cat scores')
    assert_bits(cat[2], g['cat_classes'], 'cat classes')
    host_cls, host_dl = [g['cls%d' % i] for i in range(n)], [g['box%d' % i] for i in range(n)]
    box_check.check_decode(cat[1], g['cat_boxes'], host_cls, host_dl, strides, anchors, thr, top_n, what='cat boxes')
    # (b) the batched multi-level entry produces the same bits as (a)
    fused = box.decode_levels(cls, dl, strides, thr, top_n, anchors)
    for a, b in zip(cat, fused):
        assert torch.equal(a, b)
    # (c) NMS on the REFERENCE's decoded candidates: bit-exact
    out = box.nms(cuda(g['cat_scores']), cuda(g['cat_boxes']), cuda(g['cat_classes']), nms, det)
    for o, k in zip(out, ('out_scores', 'out_boxes', 'out_classes')):
        assert_bits(o, g[k], k)
    # (d) end to end through our own decode (boxes differ by <= tolerance, which must not flip a
    #     suppression decision in these fixtures)
    e2e = box.detect(cls, dl, strides, anchors, thr, top_n, nms, det)
    assert_bits(e2e[0], g['out_scores'], 'e2e scores')
    assert_bits(e2e[2], g['out_classes'], 'e2e classes')
    ref_nms = box_oracle.nms(torch.from_numpy(g['cat_scores']), torch.from_numpy(g['cat_boxes']), torch.from_numpy(g['cat_classes']),
                             nms, det, return_indices=True)
    assert_bits(ref_nms[1], g['out_boxes'], 'oracle nms == fixture')      # so its kept positions are the reference's
    box_check.check_detections(e2e[1], g['out_boxes'], ref_nms[3], host_cls, host_dl, strides, anchors, thr, top_n,
                               what='e2e boxes')


# ------------------------------------------------------------------------------------------------
# (2) pinned oracle on fresh seeded inputs
# ------------------------------------------------------------------------------------------------
def _check_decode_levels(cls, dl, strides, anchors, thr, top_n, rotated=False):
    out = _C.decode_levels([c.cuda() for c in cls], [d.cuda() for d in dl],
                           [anchors[s][0] if rotated else anchors[s] for s in strides], strides, thr, top_n,
                           rotated, return_indices=True)
    ref = [box_oracle.decode(c, d, s, thr, top_n, anchors[s], rotated, return_indices=True)
           for c, d, s in zip(cls, dl, strides)]
    ref = [torch.cat(t, 1) for t in zip(*ref)]
    assert torch.equal(out[3].cpu().long(), ref[3]), 'selected indices / order'
    assert_bits(out[0], ref[0], 'scores')
    assert_bits(out[2], ref[2], 'classes')
    box_check.check_decode(out[1], ref[1], cls, dl, strides, anchors, thr, top_n, rotated, ref_indices=ref[3], what='boxes')
    return out, ref


@pytest.mark.parametrize('kind,seed,batch', [('sparse', 201, 2), ('dense', 202, 1), ('clustered', 203, 3)])
def test_pyramid_vs_oracle(kind, seed, batch):
    cls, dl, strides = synthetic.pyramid(batch, 9, 80, 256, 384, kind, seed)
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}
    out, ref = _check_decode_levels(cls, dl, strides, anchors, 0.05, 1000)
    # NMS on identical (oracle-decoded) candidates: indices and values bit-exact
    hip = _C.nms(ref[0].cuda(), ref[1].cuda(), ref[2].cuda(), 0.5, 100, False, return_indices=True)
    ora = box_oracle.nms(ref[0], ref[1], ref[2], 0.5, 100, return_indices=True)
    assert torch.equal(hip[3].cpu().long(), ora[3])
    for h, o in zip(hip[:3], ora[:3]):
        assert_bits(h, o, 'nms')


@pytest.mark.parametrize('shape', [(1, 3, 5, 5, 7), (2, 9, 7, 3, 3), (5, 1, 1, 1, 1), (3, 2, 3, 129, 67)],
                         ids=lambda s: 'B%d_A%d_C%d_%dx%d' % s)
def test_odd_shapes_tails_and_image_straddling_tiles(shape):
    """total % 4 != 0 (scalar tail), tiles that straddle image boundaries, tiny levels."""
    b, a, c, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    cls = synthetic.make_unique_scores(torch.rand(b, a * c, h, w, generator=g), 0.3)
    dl = torch.randn(b, a * 4, h, w, generator=g) * 0.3
    anchors = {16: box.generate_anchors(16, RATIOS, SCALES)[:a].contiguous()}
    _check_decode_levels([cls], [dl], [16], anchors, 0.3, 37)


def test_ties_follow_the_canonical_rule():
    """bf16-quantised, constant and saturated scores: massive ties.  Order must be score desc,
    flat index asc (what a stable sort gives; the reference's CUDA path behaves the same)."""
    g = torch.Generator().manual_seed(7)
    anchors = {32: box.generate_anchors(32, RATIOS, SCALES)}
    dl = torch.randn(2, 36, 13, 20, generator=g) * 0.2
    quant = (torch.randn(2, 9 * 20, 13, 20, generator=g) * 1.0 + synthetic.LOGIT_PRIOR).sigmoid().bfloat16().float()
    const = torch.full((2, 9 * 20, 13, 20), 0.25)
    sat = torch.ones(2, 9 * 20, 13, 20)
    sat[:, ::3] = 0.0
    for cls in (quant, const, sat):
        _check_decode_levels([cls], [dl], [32], anchors, 0.05, 100)
    # nms ties: equal scores, overlapping boxes -> the earlier position wins
    scores = torch.full((1, 64), 0.5)
    boxes = torch.tensor([[10., 10., 50., 50.]]).repeat(64, 1)[None] + torch.arange(64).view(1, 64, 1) * 0.25
    classes = torch.zeros(1, 64)
    hip = _C.nms(scores.cuda(), boxes.cuda(), classes.cuda(), 0.5, 10, False, return_indices=True)
    ora = box_oracle.nms(scores, boxes, classes, 0.5, 10, return_indices=True)
    assert torch.equal(hip[3].cpu().long(), ora[3])
    for h, o in zip(hip[:3], ora[:3]):
        assert_bits(h, o, 'nms ties')


def test_edge_cases():
    anchors = {8: box.generate_anchors(8, RATIOS, SCALES)}
    g = torch.Generator().manual_seed(11)
    base = synthetic.make_unique_scores(torch.rand(2, 9 * 4, 6, 10, generator=g), 0.0)
    dl = torch.randn(2, 36, 6, 10, generator=g)
    k_total = base[0].numel()
    # nothing above / everything above / K == top_n exactly / K == top_n + 1 / top_n = 1
    _check_decode_levels([base * 0.01], [dl], [8], anchors, 0.05, 50)
    _check_decode_levels([base * 0.5 + 0.5], [dl], [8], anchors, 0.05, 50)
    srt = base[0].flatten().sort(descending=True).values
    for top_n in (1, 17, 64):
        thr = float(srt[top_n - 1])                       # exactly top_n candidates in image 0 (>=)
        _check_decode_levels([base], [dl], [8], anchors, thr, top_n)
        _check_decode_levels([base], [dl], [8], anchors, float(srt[top_n]), top_n)
    assert k_total > 64
    # huge deltas: both clamps, exp overflow to inf, -inf -> all finite after the clamp
    wild = dl * 40.0
    _check_decode_levels([base], [wild], [8], anchors, 0.5, 200)
    # NaN / inf scores never pass `>=` (NaN) or sort first (inf)
    weird = base.clone()
    weird[0, 0, 0, :3] = torch.tensor([float('nan'), float('inf'), -float('inf')])
    _check_decode_levels([weird], [dl], [8], anchors, 0.5, 30)
    # negative threshold: negative scores and both zeros are candidates
    signed = base - 0.5
    signed[1, 3, 2, 2] = -0.0
    signed[1, 3, 2, 3] = 0.0
    out, ref = _check_decode_levels([signed], [dl], [8], anchors, -0.25, 400)


def test_nms_edge_cases():
    g = torch.Generator().manual_seed(13)
    # empty, single, all suppressed by the first, more survivors than detections, count not /64
    for count, ndet, n_cls in [(1, 5, 1), (63, 100, 2), (65, 3, 1), (1000, 1, 3), (4097, 100, 80), (7680, 300, 5)]:
        ctr = torch.rand(2, count, 2, generator=g) * 300
        wh = torch.rand(2, count, 2, generator=g) * 100 + 1
        boxes = torch.cat([ctr, ctr + wh], 2)
        scores = synthetic.make_unique_scores(torch.rand(2, count, generator=g), 0.0)
        scores[0, ::2] = 0.0
        scores[1] = 0.0 if count == 1 else scores[1]
        classes = torch.randint(0, n_cls, (2, count), generator=g).float()
        hip = _C.nms(scores.cuda(), boxes.cuda(), classes.cuda(), 0.5, ndet, False, return_indices=True)
        ora = box_oracle.nms(scores, boxes, classes, 0.5, ndet, return_indices=True)
        assert torch.equal(hip[3].cpu().long(), ora[3]), (count, ndet)
        for h, o in zip(hip[:3], ora[:3]):
            assert_bits(h, o, 'nms %d/%d' % (count, ndet))
    # identical boxes, same class -> exactly one survivor; different classes -> all survive
    scores = synthetic.make_unique_scores(torch.rand(1, 50, generator=g) + 0.1, 0.0)
    boxes = torch.tensor([[5., 5., 80., 90.]]).repeat(50, 1)[None]
    for classes in (torch.zeros(1, 50), torch.arange(50).float()[None]):
        hip = _C.nms(scores.cuda(), boxes.cuda(), classes.cuda(), 0.5, 100)
        ora = box_oracle.nms(scores, boxes, classes, 0.5, 100)
        for h, o in zip(hip, ora):
            assert_bits(h, o, 'identical boxes')


@pytest.mark.parametrize('batch', [1, 8, 16])
def test_batch_sizes(batch):
    cls, dl, strides = synthetic.pyramid(batch, 9, 20, 128, 192, 'dense', 300 + batch)
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}
    _check_decode_levels(cls, dl, strides, anchors, 0.05, 300)


def test_outputs_do_not_depend_on_preexisting_memory_or_workspace():
    """Outputs are fully written (no reliance on torch::zeros pre-zeroing, extensions.cpp:83-85) and
    results are reproducible call to call (atomics only decide list order, which the sort erases)."""
    cls, dl, strides = synthetic.pyramid(2, 9, 20, 128, 128, 'dense', 77)
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}
    runs = []
    for i in range(3):
        junk = torch.full((64 << 20,), float('nan'), device='cuda')   # poison the caching allocator
        del junk
        runs.append(box.detect([c.cuda() for c in cls], [d.cuda() for d in dl], strides, anchors))
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)
    assert torch.isfinite(runs[0][1]).all()


# ------------------------------------------------------------------------------------------------
# (3) the multi-workgroup selection passes (select_pass_kernel): every route through them
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('case', ['spread', 'ties-bf16', 'all-equal', 'all-equal-lists', 'one-outlier-rest-equal', 'two-plateaus', 'overflow-raw'])
def test_selection_passes_every_route(case):
    """One level of 9 x 40 x 60 x 80 = 1.73 M scores per image (14 workgroups per segment), top_n = 1000:
      spread                   two digits isolate the top 1000 (the normal route)
      ties-bf16                bf16-quantised scores: hundreds of keys tie at the boundary, resolved by index
      all-equal                every score equal (sub-lists overflow -> raw scores): every key in ONE bin, the range jump
                               starts the second digit at the first differing bit (the "saturated input" route)
      all-equal-lists          the same with 100 k equal candidates in complete lists
      one-outlier-rest-equal   pass 0 splits {1 key} / {all the rest}: the second digit cannot separate 150 k equal
                               scores, the filter pass declines (> kSurvCap survivors) and select_decode walks the lists
      two-plateaus             boundary inside a plateau of 300 k equal scores below 5 k distinct larger ones
      overflow-raw             ODTK_CAND_CAP-sized sub-lists overflow: the passes walk the raw scores"""
    g = torch.Generator().manual_seed(len(case))
    a, c, h, w = 9, 40, 60, 80
    anchors = {16: box.generate_anchors(16, RATIOS, SCALES)}
    dl = torch.randn(2, a * 4, h, w, generator=g) * 0.2
    n = a * c * h * w
    thr = 0.05
    if case == 'spread':
        cls = torch.where(torch.rand(2, a * c, h, w, generator=g) < 0.08, torch.rand(2, a * c, h, w, generator=g) * 0.9 + 0.06,
                          torch.rand(2, a * c, h, w, generator=g) * 0.04)
    elif case == 'ties-bf16':
        cls = (torch.randn(2, a * c, h, w, generator=g) + synthetic.LOGIT_PRIOR + 2.0).sigmoid().bfloat16().float()
    elif case == 'all-equal':
        cls = torch.full((2, a * c, h, w), 0.625)
    elif case == 'all-equal-lists':
        cls = torch.full((2, a * c, h, w), 0.01)
        cls.view(2, -1)[:, torch.randperm(n, generator=g)[:100000]] = 0.625
    elif case == 'one-outlier-rest-equal':
        cls = torch.full((2, a * c, h, w), 0.01)
        cls.view(2, -1)[:, torch.randperm(n, generator=g)[:150000]] = 0.25
        cls.view(2, -1)[0, 12345] = 0.75
        cls.view(2, -1)[1, n - 1] = 0.5
    elif case == 'two-plateaus':
        cls = torch.full((2, a * c, h, w), 0.01)
        flat = cls.view(2, -1)
        perm = torch.randperm(n, generator=g)
        flat[:, perm[:300000]] = 0.375
        flat[:, perm[300000:300700]] = torch.rand(2, 700, generator=g) * 0.5 + 0.4
    else:
        cls = torch.rand(2, a * c, h, w, generator=g) * 0.9 + 0.06          # every score a candidate: 1.7 M > any sub-list
    out, ref = _check_decode_levels([cls], [dl], [16], anchors, thr, 1000)
    assert int((ref[0] > 0).sum()) == 2000


# ------------------------------------------------------------------------------------------------
# (4) no caps the reference does not have: top_n = 2000 x 5 levels (10 000 NMS inputs), NMS on 24 576 candidates
# ------------------------------------------------------------------------------------------------
def test_nms_beyond_the_lds_resident_count():
    g = torch.Generator().manual_seed(29)
    for count, ndet, n_cls, rotated in [(7681, 100, 3, False), (10000, 100, 80, False), (24576, 300, 2, False), (9000, 50, 2, True)]:
        ctr = torch.rand(2, count, 2, generator=g) * 600
        wh = torch.rand(2, count, 2, generator=g) * 120 + 1
        boxes = torch.cat([ctr, ctr + wh], 2)
        if rotated:
            ang = (torch.rand(2, count, 1, generator=g) - 0.5) * 1.2
            boxes = torch.cat([boxes, torch.sin(ang), torch.cos(ang)], 2)
        scores = synthetic.make_unique_scores(torch.rand(2, count, generator=g), 0.0)
        scores[1, ::3] = 0.0
        classes = torch.randint(0, n_cls, (2, count), generator=g).float()
        hip = _C.nms(scores.cuda(), boxes.cuda(), classes.cuda(), 0.5, ndet, rotated, return_indices=True)
        if rotated:
            from oracle import c_oracle
            ora = c_oracle.nms(scores.numpy(), boxes.numpy(), classes.numpy(), 0.5, ndet, rotated=True)
            assert np.array_equal(hip[3].cpu().numpy().astype(np.int64), ora[3]), count
            for h, o in zip(hip[:3], ora[:3]):
                assert_bits(h, o, 'rotated nms %d' % count)
        else:
            ora = box_oracle.nms(scores, boxes, classes, 0.5, ndet, return_indices=True)
            assert torch.equal(hip[3].cpu().long(), ora[3]), count
            for h, o in zip(hip[:3], ora[:3]):
                assert_bits(h, o, 'nms %d/%d' % (count, ndet))


def test_top_n_2000_times_five_levels():
    """config['top_n'] = 2000 with 5 levels: 10 000 candidates per image into NMS (round 1 returned 'invalid argument')."""
    cls, dl, strides = synthetic.pyramid(2, 9, 80, 256, 384, 'dense', 515)
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}
    out, ref = _check_decode_levels(cls, dl, strides, anchors, 0.05, 2000)
    det = box.detect([c.cuda() for c in cls], [d.cuda() for d in dl], strides, anchors, 0.05, 2000, 0.5, 100)
    via = _C.nms(out[0], out[1], out[2], 0.5, 100)
    for a, b in zip(det, via):
        assert torch.equal(a, b)
    ora = box_oracle.nms(out[0].cpu(), out[1].cpu(), out[2].cpu(), 0.5, 100)
    for h, o in zip(det, ora):
        assert_bits(h, o, 'detect top_n=2000')
    assert int((ref[0] > 0).sum()) > 5000


@pytest.mark.parametrize('top_n', [4097, 5000, 16384])
def test_top_n_beyond_4096(top_n):
    """top_n > 4096 per level (round 1 / 2a: 'invalid argument'; the reference has no cap): the select_decode variant with a
    128 KiB dynamic-LDS sort buffer, through the selection passes (138 k candidates), the direct route (fewer candidates
    than top_n) and ties; NMS then takes 2 x top_n candidates per image out of the workspace."""
    g = torch.Generator().manual_seed(top_n)
    a, c, h, w = 9, 40, 60, 80
    anchors = {16: box.generate_anchors(16, RATIOS, SCALES), 32: box.generate_anchors(32, RATIOS, SCALES)}
    big = torch.where(torch.rand(2, a * c, h, w, generator=g) < 0.08, torch.rand(2, a * c, h, w, generator=g) * 0.9 + 0.06,
                      torch.rand(2, a * c, h, w, generator=g) * 0.04)
    big[1] = big[1].bfloat16().float()                                         # image 1: heavy ties
    small = torch.where(torch.rand(2, a * c, 30, 40, generator=g) < 0.005, torch.rand(2, a * c, 30, 40, generator=g) * 0.9 + 0.06,
                        torch.zeros(2, a * c, 30, 40))                         # ~2 k candidates: fewer than top_n
    dl = [torch.randn(2, a * 4, h, w, generator=g) * 0.2, torch.randn(2, a * 4, 30, 40, generator=g) * 0.2]
    out, ref = _check_decode_levels([big, small], dl, [16, 32], anchors, 0.05, top_n)
    assert int((ref[0][:, :top_n] > 0).sum()) == 2 * top_n
    det = box.detect([big.cuda(), small.cuda()], [d.cuda() for d in dl], [16, 32], anchors, 0.05, top_n, 0.5, 300)
    ora = box_oracle.nms(out[0].cpu(), out[1].cpu(), out[2].cpu(), 0.5, 300)
    for h_, o in zip(det, ora):
        assert_bits(h_, o, 'detect top_n=%d' % top_n)
