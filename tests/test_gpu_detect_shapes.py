"""`odtk_detect` (one workspace, NMS in sorted-run mode: its rounds are prefixes of decode_levels' per-level lists)
against the two-call sequence `odtk_decode_levels` + `odtk_nms` (generic mode: key list + radix selection), bit for
bit, over the run geometries the sorted-run prologue distinguishes: 1..6 runs, run lengths below / at / above a
wave, runs that are empty, full, or exhausted mid-round, tie-heavy 16-bit scores, more than 7680 candidates."""
import pytest
import torch

from odtk import _C, box, synthetic

pytestmark = pytest.mark.gpu

RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]
ALL_STRIDES = [8, 16, 32, 64, 128, 256]


def _heads(n_levels, batch, kind, dtype, seed, empty_level=None, size=(192, 256)):
    strides = ALL_STRIDES[:n_levels]
    shapes = [(max(1, size[0] // s), max(1, size[1] // s)) for s in strides]
    cls, dl = [], []
    for i, (h, w) in enumerate(shapes):
        lg, d = synthetic.make_level(batch, 9, 20, h, w, kind, seed + i, stride=strides[i], dtype=dtype)
        if i == empty_level:
            lg = torch.full_like(lg, -20.0)                            # no candidate on this level: an empty run
        cls.append(lg.cuda())
        dl.append(d.cuda())
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}
    return cls, dl, strides, anchors


@pytest.mark.parametrize('n_levels,top_n,kind,dtype,empty_level', [
    (1, 64, 'dense', torch.float32, None),            # one run of exactly one wave
    (1, 50, 'dense', torch.float32, None),            # run shorter than a wave: generic mode inside detect
    (2, 64, 'dense', torch.bfloat16, None),           # two short full runs, 16-bit ties
    (3, 129, 'dense', torch.float32, None),           # step 341: the last threads of a round belong to no run
    (3, 129, 'sparse', torch.bfloat16, 1),            # middle run empty, the others partly filled
    (5, 1000, 'dense', torch.bfloat16, 0),            # the bench geometry with an empty first run
    (6, 100, 'dense', torch.float16, None),           # step 170, six runs
    (6, 1000, 'sparse', torch.float32, 5),            # last run empty
    (5, 2000, 'dense', torch.float32, None),          # 10 000 candidates: key list in the workspace for the two-call path
    (2, 4096, 'dense', torch.float32, 1),             # runs longer than a round, one of them empty
])
@pytest.mark.parametrize('ndet,nms_thr', [(100, 0.5), (300, 0.9)])
def test_detect_equals_decode_levels_then_nms(n_levels, top_n, kind, dtype, empty_level, ndet, nms_thr):
    cls, dl, strides, anchors = _heads(n_levels, 3, kind, dtype, seed=31 * n_levels + top_n, empty_level=empty_level)
    one = box.detect(cls, dl, strides, anchors, 0.05, top_n, nms_thr, ndet, logits=True)
    dec = _C.decode_levels(cls, dl, [anchors[s] for s in strides], strides, 0.05, top_n, False, logits=True)
    two = _C.nms(dec[0], dec[1], dec[2], nms_thr, ndet)
    for a, b, what in zip(one, two, ('scores', 'boxes', 'classes')):
        assert a.shape == b.shape and torch.equal(a, b), what
    filled = (dec[0] > 0).sum(1)
    assert int(filled.max()) > 0
    if empty_level is not None:
        assert int((dec[0][:, empty_level * top_n:(empty_level + 1) * top_n] > 0).sum()) == 0
    assert int((one[0] > 0).sum(1).max()) > 0


@pytest.mark.parametrize('n_levels,top_n', [(7, 300), (10, 1000)])
def test_more_levels_than_one_level_table_holds(n_levels, top_n):
    """A model with two backbones hands over ten levels (reference model.py:138); one call's level table holds six, so
    `box.detect` decodes in groups.  Must equal the reference's own sequence: per-level decode, cat, nms."""
    cls, dl, strides, anchors = _heads(5, 2, 'dense', torch.float32, seed=77)
    more = _heads(5, 2, 'dense', torch.float32, seed=78)
    cls, dl, strides = (cls + more[0])[:n_levels], (dl + more[1])[:n_levels], (strides + more[2])[:n_levels]
    got = box.detect(cls, dl, strides, anchors, 0.05, top_n, 0.5, 100, logits=True)
    per_level = [box.decode(c.sigmoid(), d, s, 0.05, top_n, anchors[s]) for c, d, s in zip(cls, dl, strides)]
    want = box.nms(*[torch.cat(t, 1) for t in zip(*per_level)], 0.5, 100)
    for a, b, what in zip(got, want, ('scores', 'boxes', 'classes')):
        assert torch.equal(a, b), what
    assert int((got[0] > 0).sum()) > 0
