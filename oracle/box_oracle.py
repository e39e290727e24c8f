"""CPU restatement of the reference's axis-aligned post-processing (TEST INFRASTRUCTURE ONLY).

Restates, in plain torch-CPU fp32 arithmetic (the reference's own arithmetic):
  generate_anchors      <- /root/reference/odtk/box.py:8-20
  delta2box             <- /root/reference/odtk/box.py:97-111
  decode (CPU branch)   <- /root/reference/odtk/box.py:255-309
  nms    (CPU branch)   <- /root/reference/odtk/box.py:312-367

Deliberate, documented deviations from the reference text (each changes nothing on tie-free
inputs, which is what tests/golden pins bit-for-bit against the real reference):
  * integer index math uses `//` (the reference's `/` on int64 tensors only floor-divides on
    torch < 1.5; box.py:291,296,297).
  * ordering under ties is canonical: score descending, then flat NCHW index ascending
    (decode) / candidate position ascending (nms).  torch.topk (box.py:289) and
    torch.sort(stable=False) (box.py:337) return equal keys in an implementation-defined
    order; the stable rule is the one the reference's CUDA path produces with its stable
    radix sort (csrc/cuda/decode.cu:111-112, nms.cu:136-137).
  * decode can also return the selected flat indices (for index-level parity checks).

This module is the checker and the timed `cpu_baseline` ("port").  It is never imported by the
product package.
"""
import torch


def generate_anchors(stride, ratio_vals, scales_vals):
    """box.py:8-20.  Rows ordered scale-major / ratio-minor, [x1, y1, x2, y2] around one cell."""
    n_r, n_s = len(ratio_vals), len(scales_vals)
    scales = torch.tensor(scales_vals, dtype=torch.float32).repeat(n_r, 1).t().contiguous().view(-1, 1)
    ratios = torch.tensor(list(ratio_vals) * n_s, dtype=torch.float32)
    wh = torch.full((n_r * n_s, 2), float(stride), dtype=torch.float32)
    ws = torch.sqrt(wh[:, 0] * wh[:, 1] / ratios)
    dwh = torch.stack([ws, ws * ratios], dim=1)
    lo = 0.5 * (wh - dwh * scales)
    hi = 0.5 * (wh + dwh * scales)
    return torch.cat([lo, hi], dim=1)


def delta2box(deltas, anchors, size, stride):
    """box.py:97-111.  size = [W, H]; op order is normative for the HIP kernel."""
    wh = anchors[:, 2:] - anchors[:, :2] + 1
    ctr = anchors[:, :2] + 0.5 * wh
    pred_ctr = deltas[:, :2] * wh + ctr
    pred_wh = torch.exp(deltas[:, 2:]) * wh
    lo_lim = torch.zeros(2, dtype=deltas.dtype)
    hi_lim = torch.tensor([size], dtype=deltas.dtype) * stride - 1

    def clamp(t):
        return torch.max(lo_lim, torch.min(t, hi_lim))

    return torch.cat([clamp(pred_ctr - 0.5 * pred_wh), clamp(pred_ctr + 0.5 * pred_wh - 1)], 1)


def decode(all_cls_head, all_box_head, stride=1, threshold=0.05, top_n=1000, anchors=None,
           rotated=False, return_indices=False):
    """box.py:255-309 (CPU branch).  cls [B, A*C, H, W] post-sigmoid, box [B, A*nb, H, W].

    rotated=True follows csrc/cuda/decode_rotate.cu:116-167 for the gather / 6-tuple layout
    ([x1, y1, x2, y2, sin, cos], sin/cos passed through) with box.py's conventions for the
    threshold (>=), the two-sided clamp and the score-descending order;
    `anchors` is then the (axis-aligned [A,4], rotated [A,8]) pair and only [0] is used
    (box.py:258-259).
    """
    if rotated:
        anchors = anchors[0]
    nb = 6 if rotated else 4
    anchors = anchors.to(torch.float32)
    all_cls_head = all_cls_head.float()
    all_box_head = all_box_head.float()
    A = anchors.shape[0]
    B, AC, H, W = all_cls_head.shape
    C = AC // A

    out_scores = torch.zeros(B, top_n)
    out_boxes = torch.zeros(B, top_n, nb)
    out_classes = torch.zeros(B, top_n)
    out_indices = torch.full((B, top_n), -1, dtype=torch.int64)

    for b in range(B):
        cls_head = all_cls_head[b].contiguous().view(-1)
        keep = (cls_head >= threshold).nonzero().view(-1)
        if keep.numel() == 0:
            continue
        scores = cls_head[keep]
        # canonical order: score desc, flat index asc (stable sort of an index-ascending list)
        scores, order = torch.sort(scores, descending=True, stable=True)
        k = min(top_n, keep.numel())
        scores = scores[:k]
        indices = keep[order[:k]]

        classes = ((indices // W) // H) % C
        x = indices % W
        y = (indices // W) % H
        a = ((indices // C) // H) // W
        deltas = all_box_head[b].contiguous().view(A, nb, H, W)[a, :, y, x]
        grid = torch.stack([x, y, x, y], 1).to(torch.float32) * stride + anchors[a, :]
        boxes = delta2box(deltas[:, :4], grid, [W, H], stride)
        if rotated:
            boxes = torch.cat([boxes, deltas[:, 4:6]], 1)

        out_scores[b, :k] = scores
        out_boxes[b, :k] = boxes
        out_classes[b, :k] = classes.to(torch.float32)
        out_indices[b, :k] = indices

    if return_indices:
        return out_scores, out_boxes, out_classes, out_indices
    return out_scores, out_boxes, out_classes


def nms(all_scores, all_boxes, all_classes, nms=0.5, ndetections=100, return_indices=False):
    """box.py:312-367 (CPU branch): greedy class-aware NMS with the +1 pixel convention.

    Survivor test (box.py:349-351): keep j iff score_j > score_i  or  IoU(i,j) <= nms  or
    class_j != class_i.  Loop stops after `ndetections` kept boxes.
    """
    all_scores = all_scores.float()
    all_boxes = all_boxes.float()
    all_classes = all_classes.float()
    B = all_scores.shape[0]
    out_scores = torch.zeros(B, ndetections)
    out_boxes = torch.zeros(B, ndetections, 4)
    out_classes = torch.zeros(B, ndetections)
    out_indices = torch.full((B, ndetections), -1, dtype=torch.int64)

    for b in range(B):
        pos = (all_scores[b].view(-1) > 0).nonzero().view(-1)
        if pos.numel() == 0:
            continue
        scores = all_scores[b, pos]
        scores, order = torch.sort(scores, descending=True, stable=True)
        pos = pos[order]
        boxes = all_boxes[b, pos, :].view(-1, 4)
        classes = all_classes[b, pos].view(-1)
        areas = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)

        i = 0
        while i < ndetections and i < scores.numel():
            xy1 = torch.max(boxes[:, :2], boxes[i, :2])
            xy2 = torch.min(boxes[:, 2:], boxes[i, 2:])
            inter = torch.prod((xy2 - xy1 + 1).clamp(0), 1)
            criterion = ((scores > scores[i]) |
                         (inter / (areas + areas[i] - inter) <= nms) |
                         (classes != classes[i]))
            criterion[i] = True
            scores, boxes, classes, areas, pos = (t[criterion] for t in (scores, boxes, classes, areas, pos))
            i += 1

        n = min(i, scores.numel())
        out_scores[b, :n] = scores[:n]
        out_boxes[b, :n] = boxes[:n]
        out_classes[b, :n] = classes[:n]
        out_indices[b, :n] = pos[:n]

    if return_indices:
        return out_scores, out_boxes, out_classes, out_indices
    return out_scores, out_boxes, out_classes


def postprocess(cls_heads, box_heads, strides, anchors_per_stride, threshold=0.05, top_n=1000,
                nms_thresh=0.5, detections=100):
    """model.py:153-165: per-level decode -> cat over levels -> nms."""
    decoded = [decode(c, b, s, threshold, top_n, anchors_per_stride[s])
               for c, b, s in zip(cls_heads, box_heads, strides)]
    scores, boxes, classes = (torch.cat(t, 1) for t in zip(*decoded))
    return nms(scores, boxes, classes, nms_thresh, detections)


# ---------------------------------------------------------------------------------------------
# training-side target assignment
# ---------------------------------------------------------------------------------------------
def box2delta(boxes, anchors):
    """box.py:67-78."""
    a_wh = anchors[:, 2:] - anchors[:, :2] + 1
    a_ctr = anchors[:, :2] + 0.5 * a_wh
    b_wh = boxes[:, 2:] - boxes[:, :2] + 1
    b_ctr = boxes[:, :2] + 0.5 * b_wh
    return torch.cat([(b_ctr - a_ctr) / a_wh, torch.log(b_wh / a_wh)], 1)


def snap_to_anchors(boxes, size, stride, anchors, num_classes, anchor_ious):
    """box.py:134-189 restated in the anchor order [A, H, W] directly (the reference builds
    [A, W, H] and transposes; every per-anchor value is independent of that order).
    boxes [N, 5] = (x, y, w, h, class); size = [W*stride, H*stride] in pixels.
    -> cls_target [A, C, H, W], box_target [A, 4, H, W], depth [A, 1, H, W]."""
    A = anchors.shape[0]
    W, H = int(size[0] / stride), int(size[1] / stride)
    if boxes.nelement() == 0:
        return torch.zeros(A, num_classes, H, W), torch.zeros(A, 4, H, W), torch.zeros(A, 1, H, W)
    boxes, classes = boxes.split(4, dim=1)
    xs = torch.arange(0, size[0], stride, dtype=classes.dtype)
    ys = torch.arange(0, size[1], stride, dtype=classes.dtype)
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')                       # [H, W]
    grid = torch.stack((gx, gy, gx, gy), 2).unsqueeze(0)                 # [1, H, W, 4]
    anc = (grid + anchors.view(-1, 1, 1, 4).to(classes.dtype)).contiguous().view(-1, 4)   # (a, y, x)
    boxes = torch.cat([boxes[:, :2], boxes[:, :2] + boxes[:, 2:] - 1], 1)
    xy1 = torch.max(anc[:, None, :2], boxes[:, :2])
    xy2 = torch.min(anc[:, None, 2:], boxes[:, 2:])
    inter = torch.prod((xy2 - xy1 + 1).clamp(0), 2)
    boxes_area = torch.prod(boxes[:, 2:] - boxes[:, :2] + 1, 1)
    anc_area = torch.prod(anc[:, 2:] - anc[:, :2] + 1, 1)
    overlap = inter / (anc_area[:, None] + boxes_area - inter)
    overlap, indices = overlap.max(1)
    box_target = box2delta(boxes[indices], anc).view(A, H, W, 4).permute(0, 3, 1, 2).contiguous()
    depth = torch.ones_like(overlap) * -1
    depth[overlap < anchor_ious[0]] = 0
    fg = overlap >= anchor_ious[1]
    depth[fg] = classes[indices][fg].squeeze(1) + 1
    cls_idx = classes[indices].long().view(-1)
    cls_idx[overlap < anchor_ious[0]] = num_classes
    cls_target = torch.zeros(anc.shape[0], num_classes + 1, dtype=boxes.dtype)
    cls_target.scatter_(1, cls_idx.view(-1, 1), 1)
    cls_target = cls_target[:, :num_classes].view(A, H, W, num_classes).permute(0, 3, 1, 2).contiguous()
    return cls_target, box_target, depth.view(A, 1, H, W)
