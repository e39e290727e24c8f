"""Box-coordinate acceptance at the north star's tolerance (TEST INFRASTRUCTURE ONLY: tests/, smoke()).

`north_star`: boxes within 1e-4 of the reference's CPU path (reference odtk/box.py:97-111).  One operation of that path is
not bit-reproducible across implementations: exp().  torch's CPU exp -- what the reference computes with -- is off the
correctly rounded value by one ulp on ~1 % of its inputs; the HIP kernel and the C restatement round correctly.  One ulp of
exp moves `pred_wh` by an ulp and `pred_ctr -+ 0.5 * pred_wh (- 1)` rounds twice more, so a corner of a box hundreds of
pixels from the origin can land 2 ulp (1.2e-4 at 512..1024 px) away from the reference's.

The acceptance therefore keeps 1e-4 as THE bar and, for every coordinate beyond it, demands proof that the deviation is
the reference's own exp rounding and not an error of the path under test:
  (i)  the HIP box equals the C restatement's box bit for bit (same correctly rounded arithmetic), and
  (ii) the HIP box is at least as close to the float64 evaluation of box.py:97-111 as the reference's box is
       (+ 1 ulp of slack for the final rounding).
`check_boxes` returns how many coordinates needed that proof, so tests can print the count per configuration; more than
`max_proven` (2) of them in one call fail outright, proof or not -- a regression cannot hide behind the escape.  (Counted
against that bound: deviations of MORE than one fp32 ulp of the coordinate.  From 1024 px on one ulp is 1.22e-4: there the
closest any fp32 value that is not the reference's own can be is already beyond 1e-4 -- a fuzz configuration with 1500 px
boxes has half a dozen such coordinates -- so a one-ulp deviation has to pass the proof like every other, but is not a
count a regression could grow unnoticed: anything wider than the exp rounding fails proof step (ii) or the 2-ulp cap.)
"""
import numpy as np
import torch

from . import box_oracle, c_oracle

NORTH_STAR_ATOL = 1e-4


def delta2box_f64(deltas, anchors, size, stride):
    """reference odtk/box.py:97-111 evaluated in float64 on the fp32 inputs: the 'truth' the fp32 paths approximate."""
    deltas, anchors = deltas.double(), anchors.double()
    wh = anchors[:, 2:] - anchors[:, :2] + 1
    ctr = anchors[:, :2] + 0.5 * wh
    pred_ctr = deltas[:, :2] * wh + ctr
    pred_wh = torch.exp(deltas[:, 2:]) * wh
    lo = torch.zeros(2, dtype=torch.float64)
    hi = torch.tensor([size], dtype=torch.float64) * stride - 1

    def clamp(t):
        return torch.max(lo, torch.min(t, hi))

    return torch.cat([clamp(pred_ctr - 0.5 * pred_wh), clamp(pred_ctr + 0.5 * pred_wh - 1)], 1)


def truth_boxes(box_head, indices, stride, anchors, num_classes):
    """float64 boxes of the candidates with flat NCHW score indices `indices` [K] (-1 = padding -> zeros) of ONE image:
    box_head [A*4, H, W] fp32, anchors [A, 4].  Index decomposition of reference box.py:291-299."""
    anchors = anchors.to(torch.float32)
    A = anchors.shape[0]
    _, H, W = box_head.shape
    out = torch.zeros(indices.numel(), 4, dtype=torch.float64)
    valid = indices >= 0
    idx = indices[valid].long()
    x, y, a = idx % W, (idx // W) % H, ((idx // num_classes) // H) // W
    deltas = box_head.float().contiguous().view(A, 4, H, W)[a, :, y, x]
    grid = torch.stack([x, y, x, y], 1).to(torch.float32) * stride + anchors[a, :]
    out[valid] = delta2box_f64(deltas, grid, [W, H], stride)
    return out


MAX_PROVEN_PER_CALL = 2      # the escape is for the reference's exp rounding (seen 0-1 times in ~40 000 boxes): never a blanket


def check_boxes(got, ref, exact=None, truth=None, what='boxes', atol=NORTH_STAR_ATOL, max_proven=MAX_PROVEN_PER_CALL):
    """got: boxes of the path under test; ref: the reference-arithmetic (torch CPU) boxes; exact: the C restatement's boxes
    (correctly rounded exp); truth: float64 boxes (both may be callables, evaluated only when needed).  All [..., 4] (only the first four columns of rotated boxes are compared
    here: sin / cos pass through and are compared bit for bit by the callers).  Returns the number of coordinates beyond
    `atol` -- each of them proven to be exp rounding by (i) and (ii) above; anything else raises AssertionError."""
    got = torch.as_tensor(got)[..., :4].float().cpu()
    ref = torch.as_tensor(ref)[..., :4].float().cpu()
    assert got.shape == ref.shape, '%s: shapes %s vs %s' % (what, tuple(got.shape), tuple(ref.shape))
    finite = torch.isfinite(ref)
    if not bool(finite.all()):     # NaN / inf coordinates (wild deltas): the same non-finite value, nothing to measure
        assert torch.equal(torch.isfinite(got), finite) and torch.equal(torch.isnan(got), torch.isnan(ref)) and \
            torch.equal(got[~finite & ~torch.isnan(ref)], ref[~finite & ~torch.isnan(ref)]), '%s: non-finite pattern differs' % what
        got, ref = torch.where(finite, got, torch.zeros_like(got)), torch.where(finite, ref, torch.zeros_like(ref))
    diff = (got.double() - ref.double()).abs()
    over = diff > atol
    n_over = int(over.sum())
    if n_over == 0:
        return 0
    worst = float(diff[over].max())
    one_ulp = torch.from_numpy(np.spacing(ref.abs().numpy())).double()
    n_counted = int((over & (diff > one_ulp + 1e-12)).sum())   # beyond one ulp of the coordinate (below 1024 px: all of them)
    assert n_counted <= max_proven, '%s: %d coordinates beyond %g and beyond one ulp (max |diff| %.3g): more than the %d the ' \
        'exp-rounding proof may excuse per call' % (what, n_counted, atol, worst, max_proven)
    assert exact is not None and truth is not None, \
        '%s: %d coordinates beyond %g (max |diff| %.3g) and no proof inputs given' % (what, n_over, atol, worst)
    if callable(exact):        # proof inputs may be given lazily: they cost a C-oracle pass and are rarely needed
        exact = exact()
    if callable(truth):
        truth = truth()
    exact = torch.as_tensor(exact)[..., :4].float().cpu()
    truth = torch.as_tensor(truth)[..., :4].double().cpu()
    # (i) bit for bit equal to the correctly rounded restatement
    same_bits = got[over].view(torch.int32) == exact[over].view(torch.int32)
    assert bool(same_bits.all()), '%s: %d of the %d coordinates beyond %g differ from the C restatement too (max |diff| %.3g)' \
        % (what, int((~same_bits).sum()), n_over, atol, worst)
    # (ii) at least as close to the float64 truth as the reference is, + 1 ulp
    ulp = torch.from_numpy(np.spacing(ref[over].abs().numpy())).double()
    err_got, err_ref = (got[over].double() - truth[over]).abs(), (ref[over].double() - truth[over]).abs()
    closer = err_got <= err_ref + ulp
    assert bool(closer.all()), '%s: %d coordinates beyond %g are further from the float64 truth than the reference ' \
        '(|got - truth| up to %.3g vs |ref - truth| %.3g)' % (what, int((~closer).sum()), atol, float(err_got.max()), float(err_ref.max()))
    # and never more than 2 ulp of the coordinate: one ulp of exp, two more roundings
    assert bool((diff[over] <= 2 * ulp + 1e-12).all()), '%s: deviation above 2 ulp (max |diff| %.3g)' % (what, worst)
    return n_over


def reference_with_proof(cls_heads, box_heads, strides, anchors_per_stride, threshold=0.05, top_n=1000, nms_thresh=0.5,
                         detections=100):
    """The torch-CPU oracle's post-processing of reference model.py:153-165 (post-sigmoid fp32 NCHW scores, CPU tensors) plus,
    for the same detections, the C restatement's boxes and the float64 boxes:
    -> (scores [B, D], boxes [B, D, 4], classes [B, D], exact_boxes [B, D, 4], truth_boxes [B, D, 4] float64,
        candidates = dict(scores, boxes, classes, indices, exact, truth) of the decode stage, concatenated over levels)."""
    per_level, exact_l, truth_l = [], [], []
    for c, b, s in zip(cls_heads, box_heads, strides):
        c, b = c.float().contiguous(), b.float().contiguous()
        anchors = anchors_per_stride[s]
        dec = box_oracle.decode(c, b, s, threshold, top_n, anchors, return_indices=True)
        per_level.append(dec)
        cs, cb, cc, ci = c_oracle.decode(c.numpy(), b.numpy(), s, threshold, top_n, anchors.numpy())
        assert np.array_equal(ci, dec[3].numpy()), 'C restatement and torch oracle select different candidates'
        exact_l.append(torch.from_numpy(cb))
        num_classes = c.shape[1] // anchors.shape[0]
        truth_l.append(torch.stack([truth_boxes(b[i], dec[3][i], s, anchors, num_classes) for i in range(c.shape[0])]))
    scores, boxes, classes, indices = (torch.cat(t, 1) for t in zip(*per_level))
    exact, truth = torch.cat(exact_l, 1), torch.cat(truth_l, 1)
    out = box_oracle.nms(scores, boxes, classes, nms_thresh, detections, return_indices=True)
    pos = out[3].clamp(min=0)
    valid = (out[3] >= 0)[..., None]
    kept_exact = torch.gather(exact, 1, pos[..., None].expand(-1, -1, 4)) * valid
    kept_truth = torch.gather(truth, 1, pos[..., None].expand(-1, -1, 4)) * valid
    cand = dict(scores=scores, boxes=boxes, classes=classes, indices=indices, exact=exact, truth=truth)
    return out[0], out[1], out[2], kept_exact, kept_truth, cand


class ImageProof:
    """Lazy proof inputs for ONE image's decode output concatenated over the pyramid levels: `exact()` = the C restatement's
    boxes [L * top_n, 4|6], `truth()` = the float64 boxes [L * top_n, 4]; `at(positions)` gives the same pair for detections
    that are copies of candidates `positions` (what NMS emits).  Nothing is computed unless a coordinate actually exceeds the
    tolerance."""

    def __init__(self, scores, deltas, strides, anchors_per_stride, threshold, top_n, indices=None, rotated=False):
        """scores / deltas: per level [1, A*C, H, W] / [1, A*nb, H, W] fp32 CPU (post-sigmoid scores, as the reference's op
        receives them); indices: the reference's selected flat indices [L * top_n] (-1 padding) when known -- the C
        restatement's selection must then agree with them."""
        self.scores = [torch.as_tensor(s).float().contiguous() for s in scores]
        self.deltas = [torch.as_tensor(d).float().contiguous() for d in deltas]
        self.strides, self.threshold, self.top_n, self.rotated = list(strides), threshold, top_n, rotated
        self.anchors = [torch.as_tensor(anchors_per_stride[s][0] if rotated else anchors_per_stride[s]).float() for s in self.strides]
        self.indices = None if indices is None else torch.as_tensor(indices).long()
        self._exact = self._truth = None

    def exact(self):
        if self._exact is None:
            per, idx = [], []
            for s, d, st, a in zip(self.scores, self.deltas, self.strides, self.anchors):
                _, b, _, i = c_oracle.decode(s.numpy(), d.numpy(), st, self.threshold, self.top_n, a.numpy(), rotated=self.rotated)
                per.append(torch.from_numpy(b[0]))
                idx.append(torch.from_numpy(i[0]))
            idx = torch.cat(idx)
            assert self.indices is None or torch.equal(idx, self.indices), 'C restatement and reference select different candidates'
            self.indices = idx
            self._exact = torch.cat(per)
        return self._exact

    def truth(self):
        if self._truth is None:
            if self.indices is None:
                self.exact()
            per = []
            for l, (s, d, st, a) in enumerate(zip(self.scores, self.deltas, self.strides, self.anchors)):
                num_classes = s.shape[1] // a.shape[0]
                nb = 6 if self.rotated else 4
                head = d[0].view(a.shape[0], nb, d.shape[2], d.shape[3])[:, :4].reshape(a.shape[0] * 4, d.shape[2], d.shape[3])
                per.append(truth_boxes(head, self.indices[l * self.top_n:(l + 1) * self.top_n], st, a, num_classes))
            self._truth = torch.cat(per)
        return self._truth

    def at(self, positions):
        positions = torch.as_tensor(positions).long()
        pos, valid = positions.clamp(min=0), (positions >= 0)[:, None]
        return (lambda: self.exact()[pos] * valid), (lambda: self.truth()[pos] * valid)


def check_decode(got_boxes, ref_boxes, scores, deltas, strides, anchors_per_stride, threshold, top_n, rotated=False,
                 ref_indices=None, what='decode boxes'):
    """Batch form for decode outputs concatenated over levels: got / ref [B, L * top_n, 4|6]; scores / deltas per level
    [B, ...] fp32 post-sigmoid CPU tensors (or arrays).  Returns the number of coordinates beyond 1e-4 (all proven)."""
    got_boxes, ref_boxes = torch.as_tensor(got_boxes).cpu(), torch.as_tensor(ref_boxes).cpu()
    scores = [torch.as_tensor(s) for s in scores]
    deltas = [torch.as_tensor(d) for d in deltas]
    total = 0
    for b in range(got_boxes.shape[0]):
        proof = ImageProof([s[b:b + 1] for s in scores], [d[b:b + 1] for d in deltas], strides, anchors_per_stride, threshold,
                           top_n, None if ref_indices is None else torch.as_tensor(ref_indices)[b], rotated)
        total += check_boxes(got_boxes[b], ref_boxes[b], proof.exact, proof.truth, '%s, image %d' % (what, b))
    return total


def check_detections(got_boxes, ref_boxes, kept_positions, scores, deltas, strides, anchors_per_stride, threshold, top_n,
                     rotated=False, what='detections'):
    """Same for NMS outputs [B, D, 4|6] whose rows are copies of the decode candidates `kept_positions` [B, D] (-1 padding)."""
    got_boxes, ref_boxes = torch.as_tensor(got_boxes).cpu(), torch.as_tensor(ref_boxes).cpu()
    scores = [torch.as_tensor(s) for s in scores]
    deltas = [torch.as_tensor(d) for d in deltas]
    total = 0
    for b in range(got_boxes.shape[0]):
        proof = ImageProof([s[b:b + 1] for s in scores], [d[b:b + 1] for d in deltas], strides, anchors_per_stride, threshold,
                           top_n, None, rotated)
        exact, truth = proof.at(torch.as_tensor(kept_positions)[b])
        total += check_boxes(got_boxes[b], ref_boxes[b], exact, truth, '%s, image %d' % (what, b))
    return total
