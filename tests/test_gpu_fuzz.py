"""Seeded differential fuzzing of the HIP path against the oracles: random shapes, anchor/class
counts, thresholds, top_n, batch sizes, dtypes, layouts, tie densities.  Deterministic (fixed seeds) so
a failure names its case."""
import numpy as np
import pytest
import torch

from oracle import box_check, box_oracle, c_oracle
from odtk import _C, box

pytestmark = pytest.mark.gpu

RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]


def _case(seed):
    r = np.random.default_rng(seed)
    a = int(r.integers(1, 10))
    c = int(r.integers(1, 12))
    levels = int(r.integers(1, 4))
    b = int(r.integers(1, 6))
    shapes = [(int(r.integers(1, 40)), int(r.integers(1, 40))) for _ in range(levels)]
    strides = [int(r.choice([4, 8, 16, 32, 64])) for _ in range(levels)]
    thr = float(r.choice([0.0, 0.05, 0.3, 0.5, 0.9]))
    top_n = int(r.choice([1, 7, 64, 100, 1000, 1500]))
    spread = float(r.choice([0.5, 1.5, 4.0]))
    quant = str(r.choice(['none', 'bf16', 'coarse']))
    return a, c, b, shapes, strides, thr, top_n, spread, quant


@pytest.mark.parametrize('seed', range(24))
def test_decode_levels_random_configs(seed):
    a, c, b, shapes, strides, thr, top_n, spread, quant = _case(seed)
    g = torch.Generator().manual_seed(1000 + seed)
    cls, dl = [], []
    for (h, w) in shapes:
        s = (torch.randn(b, a * c, h, w, generator=g) * spread - 1.0).sigmoid()
        if quant == 'bf16':
            s = s.bfloat16().float()                       # heavy ties
        elif quant == 'coarse':
            s = (s * 8).round() / 8                        # extreme ties, exact thresholds
        cls.append(s)
        dl.append(torch.randn(b, a * 4, h, w, generator=g) * 0.5)
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES)[:a].contiguous() for s in set(strides)}
    out = _C.decode_levels([x.cuda() for x in cls], [x.cuda() for x in dl], [anchors[s] for s in strides], strides,
                           thr, top_n, False, return_indices=True)
    ref = [box_oracle.decode(x, d, s, thr, top_n, anchors[s], return_indices=True) for x, d, s in zip(cls, dl, strides)]
    ref = [torch.cat(t, 1) for t in zip(*ref)]
    assert torch.equal(out[3].cpu().long(), ref[3]), 'indices'
    assert torch.equal(out[0].cpu(), ref[0]) and torch.equal(out[2].cpu(), ref[2])
    box_check.check_decode(out[1], ref[1], cls, dl, strides, anchors, thr, top_n, ref_indices=ref[3])   # 1e-4, or proven exp rounding
    # same inputs as bf16 channels_last logits-free scores: identical selection on the rounded values
    if quant == 'bf16':
        out16 = _C.decode_levels([x.cuda().bfloat16().contiguous(memory_format=torch.channels_last) for x in cls],
                                 [x.cuda().bfloat16().contiguous(memory_format=torch.channels_last) for x in dl],
                                 [anchors[s] for s in strides], strides, thr, top_n, False, return_indices=True)
        assert torch.equal(out16[3], out[3]) and torch.equal(out16[0], out[0])


@pytest.mark.parametrize('seed', range(16))
def test_nms_random_configs(seed):
    r = np.random.default_rng(500 + seed)
    b = int(r.integers(1, 5))
    count = int(r.choice([1, 3, 64, 65, 300, 1025, 2500, 5000, 7680]))
    ndet = int(r.choice([1, 5, 100, 300]))
    thr = float(r.choice([0.0, 0.3, 0.5, 0.7, 1.0]))
    n_cls = int(r.choice([1, 2, 80]))
    rotated = bool(r.integers(0, 2)) and count <= 2500
    g = torch.Generator().manual_seed(700 + seed)
    span = float(r.choice([60.0, 300.0]))
    ctr = torch.rand(b, count, 2, generator=g) * span + 20
    wh = torch.rand(b, count, 2, generator=g) * 60 + 2
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 2)
    if rotated:
        th = (torch.rand(b, count, generator=g) - 0.5) * 3.0
        boxes = torch.cat([boxes, th.sin()[..., None], th.cos()[..., None]], 2)
    scores = torch.rand(b, count, generator=g)
    if r.integers(0, 2):
        scores = (scores * 16).round() / 16                # ties (and exact zeros)
    scores[torch.rand(b, count, generator=g) < 0.2] = 0
    classes = torch.randint(0, n_cls, (b, count), generator=g).float()
    out = _C.nms(scores.cuda(), boxes.cuda(), classes.cuda(), thr, ndet, rotated, return_indices=True)
    ref = c_oracle.nms(scores.numpy(), boxes.numpy(), classes.numpy(), thr, ndet, rotated=rotated)
    assert np.array_equal(out[3].cpu().numpy().astype(np.int64), ref[3]), 'kept positions'
    for h, e in zip(out[:3], ref[:3]):
        assert np.array_equal(np.ascontiguousarray(h.cpu().numpy()).view(np.uint32), e.view(np.uint32))
    if not rotated:
        t = box_oracle.nms(scores, boxes, classes, thr, ndet, return_indices=True)
        assert torch.equal(out[3].cpu().long(), t[3])


@pytest.mark.parametrize('seed', range(32))
def test_nms_on_detector_like_clusters_in_both_input_forms(seed):
    """Clusters of overlapping same-class boxes around a few objects, tied 16-bit scores, one to eighty classes, 64 .. 9000 candidates in
    one to six sorted runs (tools/nms_fuzz_long.py: the generator and the long run; round 6's batched pushes, the capped push over
    everything and the filter passes are taken or not depending on the case): odtk_nms_ex == the C restatement of the reference's CPU
    nms bit for bit, and odtk_nms_sorted_runs == odtk_nms_ex."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('nms_fuzz_long', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                'tools', 'nms_fuzz_long.py'))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.check_case(seed) == ''


def test_one_channel_heads_follow_their_partners_memory_format():
    """A = C = 1: the class head has ONE channel and is NCHW- and channels_last-contiguous at once; as 16-bit channels_last tensors
    the pair was refused as 'mixed formats' until round 6 (tools/decode_fuzz_long.py).  Same selection as the fp32 NCHW call."""
    g = torch.Generator().manual_seed(3)
    cls = torch.rand(2, 1, 17, 23, generator=g).bfloat16().float()
    dl = torch.randn(2, 4, 17, 23, generator=g) * 0.3
    anchors = box.generate_anchors(8, RATIOS, SCALES)[:1].contiguous()
    a = _C.decode_levels([cls.cuda()], [dl.cuda()], [anchors], [8], 0.3, 50, False, return_indices=True)
    b = _C.decode_levels([cls.cuda().bfloat16().contiguous(memory_format=torch.channels_last)],
                         [dl.cuda().bfloat16().contiguous(memory_format=torch.channels_last)], [anchors], [8], 0.3, 50, False, return_indices=True)
    assert torch.equal(a[3], b[3]) and torch.equal(a[0], b[0])
    ref = box_oracle.decode(cls, dl, 8, 0.3, 50, anchors, return_indices=True)
    assert torch.equal(a[3].cpu().long(), ref[3]) and torch.equal(a[0].cpu(), ref[0])


@pytest.mark.parametrize('shape', [(9, 20, 40, 40), (9, 80, 25, 40), (3, 7, 96, 100)])
def test_every_score_passes_on_a_mid_size_level(shape):
    """Threshold 0 on levels of 0.2-0.7 M scores: every element is a candidate, so one prefilter
    workgroup hands a whole span's worth of keys to one sub-list (sized for it; were it not, the
    raw-score fallback of select_decode would take over) -- top-n must still be the exact top-n."""
    a, c, h, w = shape
    g = torch.Generator().manual_seed(31 + h)
    cls = torch.rand(2, a * c, h, w, generator=g) * 0.999 + 0.0005
    dl = torch.randn(2, a * 4, h, w, generator=g) * 0.3
    anchors = box.generate_anchors(16, RATIOS, SCALES)[:a].contiguous()
    out = _C.decode_levels([cls.cuda()], [dl.cuda()], [anchors], [16], 0.0, 1000, False, return_indices=True)
    ref = box_oracle.decode(cls, dl, 16, 0.0, 1000, anchors, return_indices=True)
    assert torch.equal(out[0].cpu(), ref[0]), 'scores'
    # ties among fp32 uniforms are possible: compare indices where the score is unique in the reference
    uniq = torch.ones_like(ref[0], dtype=torch.bool)
    uniq[:, 1:] &= ref[0][:, 1:] != ref[0][:, :-1]
    uniq[:, :-1] &= ref[0][:, :-1] != ref[0][:, 1:]
    assert torch.equal(out[3].cpu().long()[uniq], ref[3][uniq])
    assert torch.equal(out[3].cpu().long(), ref[3]), 'indices (canonical tie order)'
