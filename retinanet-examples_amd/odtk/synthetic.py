"""Seeded synthetic head tensors for parity tests and benchmarks (no dataset / weights exist here).

Random-init RetinaNet heads score ~0.01 everywhere because of the class-prior bias
(reference odtk/model.py:115-121), i.e. zero detections, so op-level tests draw the head
tensors directly.  Distributions follow SURVEY.md section 8(d):

  sparse : cls logits ~ N(-ln 99, 0.573^2)  -> ~0.2 % of scores >= 0.05
  dense  : cls logits ~ N(-ln 99, 1.0^2)    -> ~5 %
  clustered : `sparse` plus object centres whose neighbourhood is boosted for one class,
              so NMS has real suppression work to do
  box deltas ~ N(0, 0.2^2)

Everything is generated on the CPU with an explicit torch.Generator so a (seed, shape) pair
names the same tensor everywhere.
"""
import math

import torch

LOGIT_PRIOR = -math.log(99.0)
SIGMA = {'sparse': 0.573, 'dense': 1.0}


def level_shapes(height, width, strides=(8, 16, 32, 64, 128)):
    """Feature-map (H, W) per pyramid level for a stride-128-padded input (reference fpn.py:45-61)."""
    shapes = []
    h, w = height, width
    # P3 comes from three stride-2 stages + stem; every later level halves with ceil (3x3/s2/p1 conv)
    for i, s in enumerate(strides):
        if i == 0:
            h, w = math.ceil(height / s), math.ceil(width / s)
        else:
            h, w = (h + 1) // 2, (w + 1) // 2
        shapes.append((h, w))
    return shapes


def make_level(batch, num_anchors, num_classes, height, width, kind='sparse', seed=1234,
               num_box=4, clusters=0, stride=8, dtype=torch.float32):
    """Returns (cls_logits [B, A*C, H, W], box_deltas [B, A*num_box, H, W]) on the CPU."""
    g = torch.Generator().manual_seed(seed)
    base = 'sparse' if kind == 'clustered' else kind
    logits = torch.randn(batch, num_anchors * num_classes, height, width, generator=g) * SIGMA[base] + LOGIT_PRIOR
    deltas = torch.randn(batch, num_anchors * num_box, height, width, generator=g) * 0.2
    if kind == 'clustered' or clusters:
        n = clusters or 30
        for b in range(batch):
            cy = torch.randint(0, height, (n,), generator=g)
            cx = torch.randint(0, width, (n,), generator=g)
            cc = torch.randint(0, num_classes, (n,), generator=g)
            for y, x, c in zip(cy.tolist(), cx.tolist(), cc.tolist()):
                y0, y1 = max(0, y - 1), min(height, y + 2)
                x0, x1 = max(0, x - 1), min(width, x + 2)
                for a in range(num_anchors):
                    logits[b, a * num_classes + c, y0:y1, x0:x1] += 6.0
        deltas.mul_(0.5)
    return logits.to(dtype), deltas.to(dtype)


def make_unique_scores(scores, threshold=0.0):
    """Nudge duplicate values among the entries >= threshold (per image) by whole ulps until all
    candidate scores of an image are distinct, so that any correct top-k / sort has exactly one
    answer (the reference's torch.topk / unstable torch.sort tie order is arbitrary)."""
    assert threshold >= 0.0, 'candidates must be non-negative (their bit patterns are then monotonic)'
    scores = scores.clone().contiguous()
    flat = scores.view(scores.shape[0], -1)
    for b in range(flat.shape[0]):
        row = flat[b]
        cand = (row >= threshold).nonzero().view(-1)
        if cand.numel() < 2:
            continue
        # sort ascending, then enforce strictly increasing BIT PATTERNS with the smallest bumps:
        # bits'_i = max_{j<=i}(bits_j - j) + i  (>= bits_i, and bits'_i - bits'_{i-1} >= 1)
        vals, order = torch.sort(row[cand] + 0.0, stable=True)
        bits = vals.view(torch.int32).to(torch.int64)
        ramp = torch.arange(bits.numel(), dtype=torch.int64)
        bits = torch.cummax(bits - ramp, 0).values + ramp
        row[cand[order]] = bits.to(torch.int32).view(torch.float32)
    return scores


def pyramid(batch, num_anchors, num_classes, height, width, kind='sparse', seed=1234, num_box=4,
            strides=(8, 16, 32, 64, 128), unique=True, threshold=0.05):
    """Post-sigmoid scores + deltas for all levels of a `height` x `width` input.

    With unique=True all scores >= threshold of one image are distinct ACROSS levels too (so the
    concatenated NMS input is tie-free as well)."""
    shapes = level_shapes(height, width, strides)
    cls, box = [], []
    for i, (h, w) in enumerate(shapes):
        lg, dl = make_level(batch, num_anchors, num_classes, h, w, kind, seed + i, num_box, stride=strides[i])
        cls.append(lg.sigmoid())
        box.append(dl)
    if unique:
        sizes = [c[0].numel() for c in cls]
        joint = torch.cat([c.reshape(batch, -1) for c in cls], 1)
        joint = make_unique_scores(joint, threshold)
        cls = [j.reshape(c.shape) for j, c in zip(joint.split(sizes, 1), cls)]
    return cls, box, list(strides)
