"""Detection hand-off: what happens to (scores, boxes, classes) after the post-processing kernels
(reference odtk/infer.py:72-160).

The reference appends five tensors per batch, all_gathers each of them separately, and then walks every
detection in a Python loop (`.item()` / `.tolist()` per box, reference infer.py:111-148) -- once the GPU
path takes 0.2 ms per batch that loop IS the wall time of `odtk infer`.  Here:

  * per batch ONE packed [N, D*(nb+2)+2] tensor (odtk/parallel.py), gathered with ONE all_gather at the end;
  * `detections_to_coco` converts the whole result set with array operations, producing the reference's
    wire format byte for byte: `{'image_id', 'score', 'category_id', 'bbox': [x, y, w, h(, theta)]
    (, 'segmentation': [8 corner coordinates])}`, boxes divided by the resize ratio in float32 as the
    reference does, widths `x2 - x1 + 1` in double precision exactly like its Python arithmetic on
    `.tolist()` values, duplicate image ids (DistributedSampler padding) dropped after their first
    occurrence, zero-score padding rows dropped.

Dataset reading, COCO evaluation and the CLI stay out of scope (SURVEY.md 8 / DESIGN.md 7): `infer` takes
any iterable of (images, ids, ratios) batches.
"""
import json

import numpy as np
import torch

from . import parallel


def rotated_corners(x, y, w, h, theta):
    """Corner coordinates [N, 8] = (x1,y1, x2,y2, x3,y3, x4,y4) of boxes rotated about their centre --
    the 'segmentation' polygon of a rotated detection (reference odtk/utils.py:83-101, one box at a
    time there).  float64 in, float64 out."""
    x, y, w, h, theta = (np.asarray(v, dtype=np.float64) for v in (x, y, w, h, theta))
    cx, cy = x + (w - 1) / 2, y + (h - 1) / 2
    px = np.stack([x, x, x + w - 1, x + w - 1], 1) - cx[:, None]          # corner order: (xmin,ymin), (xmin,ymax),
    py = np.stack([y, y + h - 1, y + h - 1, y], 1) - cy[:, None]          # (xmax,ymax), (xmax,ymin)
    c, s = np.cos(theta)[:, None], np.sin(theta)[:, None]
    rx = c * px - s * py + cx[:, None]
    ry = s * px + c * py + cy[:, None]
    return np.stack([rx, ry], 2).reshape(len(x), 8)


def detections_to_coco(scores, boxes, classes, ids, ratios, rotated_bbox=False, category_ids=None):
    """[N, D] scores, [N, D, 4|6] boxes, [N, D] classes, [N] image ids, [N] resize ratios (any device)
    -> list of COCO detection dicts in the reference's order (image by image, score-descending as the
    NMS wrote them).  `category_ids`: optional list mapping class index -> dataset category id
    (reference infer.py:134-135)."""
    scores = scores.detach().float().cpu()
    boxes = boxes.detach().float().cpu()
    classes = classes.detach().float().cpu()
    ids = torch.as_tensor(ids).detach().cpu().long().reshape(-1)
    ratios = torch.as_tensor(ratios).detach().float().cpu().reshape(-1)
    n, d = scores.shape
    if n == 0:
        return []

    # first occurrence of every image id, in order (reference infer.py:114-118)
    id_np = ids.numpy()
    _, first = np.unique(id_np, return_index=True)
    image_rows = np.zeros(n, dtype=bool)
    image_rows[first] = True

    # same float32 division as the reference (`boxes / ratios`), before anything becomes a double
    scaled = boxes.clone()
    scaled[:, :, :4] = scaled[:, :, :4] / ratios.view(n, 1, 1)

    keep = (scores > 0) & torch.from_numpy(image_rows).view(n, 1)         # padding rows have score 0
    row, col = keep.nonzero(as_tuple=True)                                  # row-major: image by image
    if row.numel() == 0:
        return []
    s = scores[row, col].numpy().astype(np.float64)
    b = scaled[row, col].numpy().astype(np.float64)
    cat = classes[row, col].to(torch.int32).numpy()                          # `.int()` truncation, as the reference
    if category_ids is not None:
        cat = np.asarray(category_ids)[cat]
    img = id_np[row.numpy()]
    x1, y1 = b[:, 0], b[:, 1]
    w, h = b[:, 2] - b[:, 0] + 1, b[:, 3] - b[:, 1] + 1
    if rotated_bbox:
        theta = np.arctan2(b[:, 4], b[:, 5])
        bbox = np.stack([x1, y1, w, h, theta], 1).tolist()
        seg = rotated_corners(x1, y1, w, h, theta).tolist()
    else:
        bbox = np.stack([x1, y1, w, h], 1).tolist()
        seg = None
    img, s, cat = img.tolist(), s.tolist(), cat.tolist()
    out = []
    for i in range(len(s)):
        det = {'image_id': img[i], 'score': s[i], 'category_id': cat[i], 'bbox': bbox[i]}
        if seg is not None:
            det['segmentation'] = [seg[i]]
        out.append(det)
    return out


def infer(model, batches, rotated_bbox=False, category_ids=None, detections_file=None, dataset=None):
    """Run `model` (eval-mode Model or FusedRetinaNet) over `batches` = iterable of
    (images [B,3,H,W] on the model's device, ids [B], ratios [B]); gather every rank's detections with
    one collective; on rank 0 convert them and optionally write the reference's JSON document
    (`{'annotations': [...], 'images': ..., 'categories': ...}`, reference infer.py:150-158).
    Returns the detection list on rank 0, None elsewhere."""
    packed = []
    detections_per_image, nb = None, 6 if rotated_bbox else 4
    with torch.no_grad():
        for images, ids, ratios in batches:
            images = images.contiguous(memory_format=torch.channels_last)
            scores, boxes, classes = model(images)
            detections_per_image = scores.shape[1]
            ids = torch.as_tensor(ids, device=scores.device)
            ratios = torch.as_tensor(ratios, device=scores.device)
            packed.append(parallel.pack_detections(scores.float(), boxes.float(), classes.float(), ids, ratios))
    if not packed:
        return [] if parallel.is_master() else None
    everything = parallel.gather_packed(torch.cat(packed, 0))
    if not parallel.is_master():
        return None
    detections = detections_to_coco(*parallel.unpack_detections(everything, detections_per_image, nb),
                                    rotated_bbox=rotated_bbox, category_ids=category_ids)
    if detections_file and detections:
        doc = {'annotations': detections}
        if dataset is not None:
            doc['images'] = dataset['images']
            if 'categories' in dataset:
                doc['categories'] = dataset['categories']
        for path in ([detections_file] if isinstance(detections_file, str) else detections_file):
            with open(path, 'w') as f:
                json.dump(doc, f, indent=4)
    return detections
