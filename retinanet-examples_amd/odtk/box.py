"""odtk.box -- Python surface of the per-anchor post-processing path, MI355X edition.

Same function names, argument order and return conventions as the reference's odtk/box.py
(generate_anchors :8, generate_anchors_rotated :23, box2delta :67, delta2box :97, decode :255,
nms :312, nms_rotated :370), so `odtk.model.Model.forward` and user code keep calling
`decode(cls, box, stride, threshold, top_n, anchors, rotated)` and `nms(scores, boxes, classes,
nms, ndetections)` unchanged.

Differences, all deliberate:
  * GPU tensors ALWAYS run the hand-written HIP kernels (libodtk_hip.so via odtk._C); a missing
    library raises, nothing falls back silently.
  * CPU tensors take the pure-torch branch below (`_decode_cpu`, `_nms_cpu`) -- the reference's own
    "no GPU" plumbing path (box.py:266-309, :319-367; BASELINE config 0), with the integer divisions
    written as floor divisions (the reference's `/` on index tensors broke with torch >= 1.5) and the
    canonical tie order (stable sorts).  The reference dispatches on `torch.cuda.is_available()`;
    here the tensor's device decides.  Rotated boxes have no CPU path (the reference's is dead code:
    box.py:408 calls an undefined `iou`).
  * `decode_levels` / `detect` are additions: all pyramid levels x the whole batch in one
    enqueue, with no host synchronisation (GPU only).
"""

import torch

from . import _C


def _grid_of_shapes(ratio_vals, scales_vals):
    """(scale, ratio) pairs in the reference's order: scale-major, ratio-minor (box.py:11-13)."""
    scales = torch.tensor([s for s in scales_vals for _ in ratio_vals], dtype=torch.float32).view(-1, 1)
    ratios = torch.tensor([r for _ in scales_vals for r in ratio_vals], dtype=torch.float32)
    return scales, ratios


def generate_anchors(stride, ratio_vals, scales_vals, angles_vals=None):
    """Base anchors [A, 4] = [x1, y1, x2, y2] around one stride x stride cell (reference box.py:8-20)."""
    scales, ratios = _grid_of_shapes(ratio_vals, scales_vals)
    cell = torch.full((ratios.numel(), 2), float(stride), dtype=torch.float32)
    side = torch.sqrt(cell[:, 0] * cell[:, 1] / ratios)
    extent = torch.stack([side, side * ratios], dim=1) * scales
    return torch.cat([0.5 * (cell - extent), 0.5 * (cell + extent)], dim=1)


def _order_quads(quads):
    """[Q, 4, 2] corner sets -> [tl, tr, br, bl] per quad (reference utils.py:15-31), batched."""
    by_x = torch.gather(quads, 1, torch.argsort(quads[:, :, 0], dim=1)[:, :, None].expand(-1, -1, 2))
    left, right = by_x[:, :2], by_x[:, 2:]
    left = torch.gather(left, 1, torch.argsort(left[:, :, 1], dim=1)[:, :, None].expand(-1, -1, 2))
    tl, bl = left[:, 0], left[:, 1]
    dist = torch.cdist(tl[:, None, :], right)[:, 0]                       # [Q, 2]
    far_first = torch.argsort(dist, dim=1, descending=True)
    right = torch.gather(right, 1, far_first[:, :, None].expand(-1, -1, 2))
    br, tr = right[:, 0], right[:, 1]
    return torch.stack([tl, tr, br, bl], dim=1)


def generate_anchors_rotated(stride, ratio_vals, scales_vals, angles_vals):
    """(axis-aligned [A, 4], rotated corner [A, 8]) anchors, angle-major (reference box.py:23-64).
    decode only consumes [0] (box.py:258-259); [1] feeds rotated target assignment."""
    scales, ratios = _grid_of_shapes(ratio_vals, scales_vals)
    n_shapes = ratios.numel()
    cell = torch.full((n_shapes, 2), float(stride), dtype=torch.float32)
    side = torch.round(torch.sqrt(cell[:, 0] * cell[:, 1] / ratios))
    extent = torch.stack([side, torch.round(side * ratios)], dim=1) * scales
    p0 = 0.5 * (cell - extent)
    p2 = 0.5 * (cell + extent) - 1
    span = p2 - p0
    p1 = p0 + span * torch.tensor([0.0, 1.0])
    p3 = p0 + span * torch.tensor([1.0, 0.0])

    angles = torch.tensor(angles_vals, dtype=torch.float32)
    n_ang = angles.numel()
    rot = torch.stack([torch.stack([torch.cos(angles), torch.sin(angles)], dim=1),
                       torch.stack([-torch.sin(angles), torch.cos(angles)], dim=1)], dim=1)   # [n_ang, 2, 2]
    half = stride / 2

    def spin(p):   # rotate about the cell centre (box.py:50-53), result [n_ang * n_shapes, 2]
        r = torch.matmul(rot, p.transpose(1, 0) - half + 0.5) + half - 0.5
        return r.permute(0, 2, 1).contiguous().view(-1, 2)

    axis = torch.cat([p0.repeat(n_ang, 1), p2.repeat(n_ang, 1)], dim=1)
    corners = _order_quads(torch.stack([spin(p0), spin(p1), spin(p2), spin(p3)], dim=1)).view(-1, 8)
    return axis, corners


def _split_anchor(anchors):
    size = anchors[:, 2:4] - anchors[:, :2] + 1
    return size, anchors[:, :2] + 0.5 * size


def box2delta(boxes, anchors):
    """Regression targets of `boxes` w.r.t. `anchors` (+1 pixel convention; reference box.py:67-78)."""
    a_size, a_ctr = _split_anchor(anchors)
    b_size, b_ctr = _split_anchor(boxes)
    return torch.cat([(b_ctr - a_ctr) / a_size, torch.log(b_size / a_size)], 1)


def box2delta_rotated(boxes, anchors):
    """As box2delta plus pass-through (sin, cos) columns (reference box.py:81-94)."""
    return torch.cat([box2delta(boxes[:, :4], anchors[:, :4]), boxes[:, 4:6]], 1)


def delta2box(deltas, anchors, size, stride):
    """Inverse of box2delta with the two-sided clamp to [0, size*stride-1] (reference box.py:97-111).
    Pure torch; the inference path does this inside the HIP decode kernel instead."""
    a_size, a_ctr = _split_anchor(anchors)
    ctr = deltas[:, :2] * a_size + a_ctr
    ext = torch.exp(deltas[:, 2:4]) * a_size
    upper = torch.tensor([size], device=deltas.device, dtype=deltas.dtype) * stride - 1
    lower = torch.zeros(2, device=deltas.device, dtype=deltas.dtype)
    lo = torch.max(lower, torch.min(ctr - 0.5 * ext, upper))
    hi = torch.max(lower, torch.min(ctr + 0.5 * ext - 1, upper))
    return torch.cat([lo, hi], 1)


def delta2box_rotated(deltas, anchors, size, stride):
    """delta2box + theta = atan2(sin, cos) (reference box.py:114-131)."""
    return torch.cat([delta2box(deltas[:, :4], anchors[:, :4], size, stride),
                      torch.atan2(deltas[:, 4], deltas[:, 5])[:, None]], 1)


def _cell_anchors(anchors, size, stride, dtype, device):
    """All anchors of a level in (anchor, y, x) order, [A*H*W, K] (K = 4 corners or 8 quad coords)."""
    xs = torch.arange(0, size[0], stride, device=device, dtype=dtype)
    ys = torch.arange(0, size[1], stride, device=device, dtype=dtype)
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')
    k = anchors.shape[1]
    grid = torch.stack([gx, gy] * (k // 2), 2).unsqueeze(0)                       # [1, H, W, K]
    return (grid + anchors.view(-1, 1, 1, k).to(device=device, dtype=dtype)).reshape(-1, k)


def _targets_from_overlap(overlap, boxes_xyxy, classes, cell, num_anchors, height, width, num_classes, anchor_ious,
                          to_delta):
    """Shared tail of snap_to_anchors[_rotated] (reference box.py:165-189 / :228-252): best box per
    anchor -> regression target, depth (-1 ignore / 0 background / class+1) and one-hot class target,
    produced directly in [A, *, H, W] order."""
    overlap, best = overlap.max(1)
    box_target = to_delta(boxes_xyxy[best], cell)
    nb = box_target.shape[1]
    box_target = box_target.view(num_anchors, height, width, nb).permute(0, 3, 1, 2).contiguous()
    background = overlap < anchor_ious[0]
    foreground = overlap >= anchor_ious[1]
    best_cls = classes[best].view(-1)
    depth = torch.full_like(overlap, -1)
    depth[background] = 0
    depth[foreground] = best_cls[foreground] + 1
    onehot_idx = best_cls.long()
    onehot_idx[background] = num_classes                                          # background has no class
    cls_target = torch.zeros((cell.shape[0], num_classes + 1), device=overlap.device, dtype=boxes_xyxy.dtype)
    cls_target.scatter_(1, onehot_idx.view(-1, 1), 1)
    cls_target = cls_target[:, :num_classes].view(num_anchors, height, width, num_classes).permute(0, 3, 1, 2)
    return cls_target.contiguous(), box_target, depth.view(num_anchors, 1, height, width)


def snap_to_anchors(boxes, size, stride, anchors, num_classes, device, anchor_ious):
    """Training targets for one image and one pyramid level (reference box.py:134-189).

    boxes [N, 5] = (x, y, w, h, class); size = [W*stride, H*stride].  Returns
    cls_target [A, C, H, W], box_target [A, 4, H, W], depth [A, 1, H, W].  Pure torch like the
    reference's (which has no native op here) and device-agnostic; per-anchor values equal the
    reference's (it builds [A, W, H] and transposes, which changes no value)."""
    num_anchors = anchors.shape[0]
    width, height = int(size[0] / stride), int(size[1] / stride)
    if boxes.nelement() == 0:
        return (torch.zeros([num_anchors, num_classes, height, width], device=device),
                torch.zeros([num_anchors, 4, height, width], device=device),
                torch.zeros([num_anchors, 1, height, width], device=device))
    boxes, classes = boxes.split(4, dim=1)
    cell = _cell_anchors(anchors, size, stride, classes.dtype, device)
    xyxy = torch.cat([boxes[:, :2], boxes[:, :2] + boxes[:, 2:] - 1], 1)
    lo = torch.max(cell[:, None, :2], xyxy[:, :2])
    hi = torch.min(cell[:, None, 2:], xyxy[:, 2:])
    inter = torch.prod((hi - lo + 1).clamp(0), 2)
    area_b = torch.prod(xyxy[:, 2:] - xyxy[:, :2] + 1, 1)
    area_a = torch.prod(cell[:, 2:] - cell[:, :2] + 1, 1)
    overlap = inter / (area_a[:, None] + area_b - inter)
    return _targets_from_overlap(overlap, xyxy, classes, cell, num_anchors, height, width, num_classes, anchor_ious,
                                 box2delta)


def snap_to_anchors_batched(targets, width, height, stride, anchors, num_classes, anchor_ious, want_cls_target=True):
    """snap_to_anchors for every image of the batch in ONE fused HIP launch (GPU only).
    targets [B, N, 5] padded with class = -1 rows (reference data.py:154-161 format).
    Returns the stacked (cls_target [B,A,C,H,W], box_target [B,A,4,H,W], depth [B,A,1,H,W])."""
    _require_gpu(targets, 'snap_to_anchors_batched')
    return _C.snap_to_anchors(targets.float().contiguous(), anchors, num_classes, int(height), int(width),
                              int(stride), anchor_ious[0], anchor_ious[1], want_cls_target)


MAX_LEVELS_PER_CALL = _C.MAX_LEVELS


def snap_to_anchors_levels(targets, sizes, strides, anchors_list, num_classes, anchor_ious, want_cls_target=True):
    """snap_to_anchors_batched for every pyramid level in ONE fused HIP launch (GPU only; csrc/targets.hpp).
    sizes: per level (H, W) of the head tensors; -> lists (cls_targets | Nones, box_targets, depths)."""
    _require_gpu(targets, 'snap_to_anchors_levels')
    return _C.snap_to_anchors_levels(targets.float().contiguous(), anchors_list, num_classes, [(int(h), int(w)) for h, w in sizes],
                                     [int(s) for s in strides], anchor_ious[0], anchor_ious[1], want_cls_target)


def rotate_boxes(boxes, points=False):
    """(x, y, w, h, theta) targets -> ([x1, y1, x2, y2, sin, cos], ordered corner quads [N, 8])
    (reference utils.py:33-82; `points=True` takes (x1, y1, x2, y2, theta))."""
    theta = boxes[:, 4]
    cos, sin = torch.cos(theta), torch.sin(theta)
    if points:
        x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
        cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
    else:
        x1, y1 = boxes[:, 0], boxes[:, 1]
        x2, y2 = x1 + boxes[:, 2], y1 + boxes[:, 3]
        cx, cy = x1 + boxes[:, 2] / 2, y1 + boxes[:, 3] / 2
    quads = []
    for px, py in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)):
        dx, dy = px - cx, py - cy
        quads.append(torch.stack([cos * dx + sin * dy + cx, -sin * dx + cos * dy + cy], 1))
    axis = torch.cat([boxes[:, :2], boxes[:, :2] + boxes[:, 2:4] - 1, sin[:, None], cos[:, None]], 1)
    return axis, _order_quads(torch.stack(quads, 1)).view(-1, 8)


def snap_to_anchors_rotated(boxes, size, stride, anchors, num_classes, device, anchor_ious):
    """Rotated training targets (reference box.py:192-252): overlap = polygon IoU between the
    ground-truth quads and the rotated anchor quads, computed by the HIP `iou` op (GPU only -- the
    reference has no CPU implementation of it either, box.py:220-223)."""
    anchors_axis, anchors_rotated = anchors
    num_anchors = anchors_rotated.shape[0]
    width, height = int(size[0] / stride), int(size[1] / stride)
    if boxes.nelement() == 0:
        return (torch.zeros([num_anchors, num_classes, height, width], device=device),
                torch.zeros([num_anchors, 6, height, width], device=device),
                torch.zeros([num_anchors, 1, height, width], device=device))
    boxes, classes = boxes.split(5, dim=1)
    boxes_axis, boxes_quads = rotate_boxes(boxes)
    boxes_axis, boxes_quads = boxes_axis.to(device), boxes_quads.to(device)
    cell_axis = _cell_anchors(anchors_axis, size, stride, torch.float32, device)
    cell_quads = _cell_anchors(anchors_rotated, size, stride, torch.float32, device)
    _require_gpu(boxes_quads, 'snap_to_anchors_rotated')
    overlap = _C.iou(boxes_quads.contiguous().view(-1), cell_quads.contiguous().view(-1))[0]
    return _targets_from_overlap(overlap, boxes_axis, classes, cell_axis, num_anchors, height, width, num_classes,
                                 anchor_ious, box2delta_rotated)


_ROTATED_ANCHOR_TABLES = {}


def snap_to_anchors_rotated_levels(targets, sizes, strides, anchors_list, num_classes, anchor_ious, want_cls_target=True):
    """snap_to_anchors_rotated for every image and every pyramid level in ONE fused HIP launch (GPU only; csrc/targets.hpp):
    no [27*H*W, N] IoU matrix, no per-image loop.  targets [B, N, 6] = (x, y, w, h, theta, class), class = -1 rows are padding
    (reference data.py format); anchors_list: per level the (axis, rotated) pair of generate_anchors_rotated; sizes: per level
    (H, W) of the head tensors.  The few per-box torch ops of the reference's `rotate_boxes` (cos / sin, corner ordering) run
    once for the whole batch.  -> lists (cls_targets | Nones, box_targets [B, A, 6, H, W], depths [B, A, 1, H, W])."""
    _require_gpu(targets, 'snap_to_anchors_rotated_levels')
    targets = targets.float()
    b, n, _ = targets.shape
    flat = targets.reshape(b * n, 6)
    axis, quads = rotate_boxes(flat[:, :5])
    tables = []
    for (anchors_axis, anchors_rotated), s in zip(anchors_list, strides):
        key = (targets.device, anchors_axis.data_ptr(), anchors_rotated.data_ptr())
        hit = _ROTATED_ANCHOR_TABLES.get(key)
        if hit is None:                                           # device copies of the (tiny) anchor tables, made once
            hit = (anchors_axis.to(targets.device, torch.float32).contiguous(), anchors_rotated.to(targets.device, torch.float32).contiguous(),
                   anchors_axis, anchors_rotated)                  # (the host tensors are kept alive: their addresses are the key)
            _ROTATED_ANCHOR_TABLES[key] = hit
        tables.append(hit[:2])
    return _C.snap_to_anchors_rotated_levels(axis.view(b, n, 6).contiguous(), quads.view(b, n, 8).contiguous(), flat[:, 5].reshape(b, n).contiguous(),
                                             tables, num_classes, [(int(h), int(w)) for h, w in sizes], [int(s) for s in strides],
                                             anchor_ious[0], anchor_ious[1], want_cls_target)


def _require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError('odtk.box.%s: tensors must be on the GPU (only the axis-aligned decode / nms have a '
                           'pure-torch CPU branch)' % what)


def _decode_cpu(cls_heads, box_heads, stride, threshold, top_n, anchors):
    """CPU branch of `decode` (reference box.py:266-309): per image keep `score >= threshold`, take the
    `top_n` best in (score desc, flat NCHW index asc) order, decode their boxes with `delta2box`."""
    cls_heads, box_heads = cls_heads.float(), box_heads.float()
    anchors = anchors.to(cls_heads.device, torch.float32)
    batch, channels, height, width = cls_heads.shape
    num_anchors = anchors.shape[0]
    num_classes = channels // num_anchors
    scores_out = cls_heads.new_zeros((batch, top_n))
    boxes_out = cls_heads.new_zeros((batch, top_n, 4))
    classes_out = cls_heads.new_zeros((batch, top_n))
    flat_scores = cls_heads.reshape(batch, -1)
    deltas_by_cell = box_heads.reshape(batch, num_anchors, 4, height, width)
    for image in range(batch):
        row = flat_scores[image]
        candidates = torch.nonzero(row >= threshold).view(-1)                 # ascending flat index
        if candidates.numel() == 0:
            continue
        ranked = torch.argsort(row[candidates], descending=True, stable=True)[:top_n]
        index = candidates[ranked]
        k = index.numel()
        a, c, y, x = torch.unravel_index(index, (num_anchors, num_classes, height, width))
        cell = torch.stack([x, y, x, y], 1).to(torch.float32) * stride + anchors[a]
        scores_out[image, :k] = row[index]
        boxes_out[image, :k] = delta2box(deltas_by_cell[image, a, :, y, x], cell, [width, height], stride)
        classes_out[image, :k] = c.to(torch.float32)
    return scores_out, boxes_out, classes_out


def _nms_cpu(all_scores, all_boxes, all_classes, nms, ndetections):
    """CPU branch of `nms` (reference box.py:319-367): greedy, class-aware, +1 pixel IoU.  The reference
    compacts its arrays after every kept box; here one `alive` mask over the score-sorted candidates does
    the same bookkeeping: the i-th kept box is the i-th alive entry, and it retires every alive entry that
    fails the reference's survivor test `score > s_i  or  IoU <= nms  or  class != c_i`."""
    all_scores, all_boxes, all_classes = all_scores.float(), all_boxes.float(), all_classes.float()
    batch = all_scores.shape[0]
    scores_out = all_scores.new_zeros((batch, ndetections))
    boxes_out = all_scores.new_zeros((batch, ndetections, 4))
    classes_out = all_scores.new_zeros((batch, ndetections))
    for image in range(batch):
        positive = torch.nonzero(all_scores[image] > 0).view(-1)
        if positive.numel() == 0:
            continue
        order = positive[torch.argsort(all_scores[image, positive], descending=True, stable=True)]
        scores, boxes, classes = all_scores[image, order], all_boxes[image, order], all_classes[image, order]
        area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
        alive = torch.ones_like(scores, dtype=torch.bool)
        kept = 0
        while kept < ndetections:
            remaining = torch.nonzero(alive).view(-1)
            if kept >= remaining.numel():
                break
            i = int(remaining[kept])
            lo = torch.max(boxes[:, :2], boxes[i, :2])
            hi = torch.min(boxes[:, 2:], boxes[i, 2:])
            inter = torch.prod((hi - lo + 1).clamp(0), 1)
            survives = (scores > scores[i]) | (inter / (area + area[i] - inter) <= nms) | (classes != classes[i])
            survives[i] = True
            alive &= survives
            kept += 1
        winners = torch.nonzero(alive).view(-1)[:kept]
        scores_out[image, :kept] = scores[winners]
        boxes_out[image, :kept] = boxes[winners]
        classes_out[image, :kept] = classes[winners]
    return scores_out, boxes_out, classes_out


def decode(all_cls_head, all_box_head, stride=1, threshold=0.05, top_n=1000, anchors=None, rotated=False):
    """Box decoding and filtering for one pyramid level (reference box.py:255-309).

    cls [B, A*C, H, W] post-sigmoid, box [B, A*{4|6}, H, W] -> scores [B, top_n],
    boxes [B, top_n, {4|6}], classes [B, top_n] (score-descending, zero padded)."""
    if rotated:
        anchors = anchors[0]
    if not all_cls_head.is_cuda:
        if rotated:
            _require_gpu(all_cls_head, 'decode(rotated=True)')
        return _decode_cpu(all_cls_head, all_box_head, stride, threshold, top_n, anchors)
    return _C.decode(all_cls_head.float().contiguous(), all_box_head.float().contiguous(),
                     anchors.reshape(-1).tolist(), stride, threshold, top_n, rotated)


def nms(all_scores, all_boxes, all_classes, nms=0.5, ndetections=100):
    """Batched class-aware greedy NMS (reference box.py:312-367)."""
    if not all_scores.is_cuda:
        return _nms_cpu(all_scores, all_boxes, all_classes, nms, ndetections)
    return _C.nms(all_scores.float().contiguous(), all_boxes.float().contiguous(),
                  all_classes.float().contiguous(), nms, ndetections, False)


def nms_rotated(all_scores, all_boxes, all_classes, nms=0.5, ndetections=100):
    """NMS on [x1, y1, x2, y2, sin, cos] boxes with polygon IoU (reference box.py:370-375)."""
    _require_gpu(all_scores, 'nms_rotated')
    return _C.nms(all_scores.float().contiguous(), all_boxes.float().contiguous(),
                  all_classes.float().contiguous(), nms, ndetections, True)


def _as_written(t):
    """Hand a head tensor to the HIP path as the convolution wrote it: float32 / bfloat16 / float16 in
    NCHW or channels_last need NO copy; anything else is converted once."""
    if t.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        t = t.float()
    if not (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)):
        t = t.contiguous()
    return t


def _pair(c, b):
    c, b = _as_written(c), _as_written(b)
    if b.dtype != c.dtype:
        b = b.to(c.dtype)
    if c.is_contiguous() != b.is_contiguous():      # one memory format per level
        b = b.contiguous() if c.is_contiguous() else b.contiguous(memory_format=torch.channels_last)
    return c, b


def decode_levels(cls_heads, box_heads, strides, threshold, top_n, anchors_per_stride, rotated=False, logits=False):
    """All levels at once: equals `[torch.cat(t, 1) for t in zip(*[decode(...) per level])]`
    (reference model.py:153-164) in one enqueue.  logits=True fuses the sigmoid (model.py:140)."""
    anchors = [anchors_per_stride[s][0] if rotated else anchors_per_stride[s] for s in strides]
    for t in cls_heads:
        _require_gpu(t, 'decode_levels')
    pairs = [_pair(c, b) for c, b in zip(cls_heads, box_heads)]
    return _C.decode_levels([p[0] for p in pairs], [p[1] for p in pairs], anchors, strides, threshold, top_n,
                            rotated, logits=logits)


def detect(cls_heads, box_heads, strides, anchors_per_stride, threshold=0.05, top_n=1000, nms=0.5,
           ndetections=100, rotated=False, logits=False, cls_bias=None, box_bias=None, cls_thresholds=None):
    """sigmoid (logits=True) + decode of all levels + nms: the whole inference post-processing of the
    reference (model.py:140-165) in one enqueue of three launches (rotated: five to seven), reading the head tensors in place.
    cls_bias / box_bias: the heads' last-conv biases, added inside the kernels (see _C.decode_levels); cls_thresholds: the
    prefilter's threshold table for that cls_bias, made once by _C.prefilter_thresholds (optional)."""
    anchors = [anchors_per_stride[s][0] if rotated else anchors_per_stride[s] for s in strides]
    for t in cls_heads:
        _require_gpu(t, 'detect')
    pairs = [_pair(c, b) for c, b in zip(cls_heads, box_heads)]
    if len(pairs) > _C.MAX_LEVELS:
        # a model with several backbones has 5 levels per backbone (reference model.py:138): the level table of one call
        # holds MAX_LEVELS, so decode in groups and hand the concatenation to nms (its candidate count is not capped)
        def bias_of(bias, lo, hi):
            return bias[lo:hi] if isinstance(bias, (list, tuple)) else bias
        parts = []
        for lo in range(0, len(pairs), _C.MAX_LEVELS):
            hi = min(lo + _C.MAX_LEVELS, len(pairs))
            parts.append(_C.decode_levels([p[0] for p in pairs[lo:hi]], [p[1] for p in pairs[lo:hi]], anchors[lo:hi],
                                          strides[lo:hi], threshold, top_n, rotated, logits=logits,
                                          cls_bias=bias_of(cls_bias, lo, hi), box_bias=bias_of(box_bias, lo, hi)))
        return _C.nms(*[torch.cat(t, 1) for t in zip(*parts)], nms, ndetections, rotated)
    return _C.detect([p[0] for p in pairs], [p[1] for p in pairs], anchors, strides, threshold, top_n, nms,
                     ndetections, rotated, logits=logits, cls_bias=cls_bias, box_bias=box_bias, cls_thresholds=cls_thresholds)
