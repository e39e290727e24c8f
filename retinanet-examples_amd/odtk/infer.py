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

`infer_batches` takes any iterable of (images, ids, ratios) batches; `infer` is the reference's entry point
(same positional arguments, infer.py:18-19): images under `path` (+ optional COCO annotations) through
odtk/data.py, the loop above, then -- when annotations with ground truth were given -- AP / AR from
odtk/cocoeval.py in place of pycocotools.
"""
import json
import os
import time

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


def infer_batches(model, batches, rotated_bbox=False, category_ids=None, detections_file=None, dataset=None,
                  on_batch=None):
    """Run `model` (eval-mode Model or FusedRetinaNet) over `batches` = iterable of
    (images [B,3,H,W] on the model's device, ids [B], ratios [B]); gather every rank's detections with
    one collective; on rank 0 convert them and optionally write the reference's JSON document
    (`{'annotations': [...], 'images': ..., 'categories': ...}`, reference infer.py:150-158).
    Returns the detection list on rank 0, None elsewhere.  `on_batch(i, forward_seconds)`: progress hook."""
    packed = []
    detections_per_image, nb = None, 6 if rotated_bbox else 4
    with torch.no_grad():
        for i, (images, ids, ratios) in enumerate(batches):
            images = images.contiguous(memory_format=torch.channels_last)
            started = time.time()
            scores, boxes, classes = model(images)
            if on_batch is not None:
                on_batch(i, time.time() - started)
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


def infer(model, path, detections_file, resize, max_size, batch_size, mixed_precision=True, is_master=True, world=0,
          annotations=None, with_apex=False, use_dali=False, is_validation=False, verbose=True, rotated_bbox=False,
          num_workers=2):
    """Run inference on the images under `path` -- the reference's `infer.infer` (infer.py:18-177), same
    arguments.  Returns `COCOeval.stats` (mAP first) when ground truth was given, None when nothing was
    detected, 0 otherwise (and on the ranks other than the master), as the reference does.

    `mixed_precision` on a GPU = fp16 autocast, like the reference's (apex O2 / torch.cuda.amp, infer.py:54-58), which
    routes `Model.forward` to the fp16 BN-folded engine (measured AP against the fp32 pipeline: 0.995-1.000, the bf16
    engine 0.963-0.988; profiles/r04_bf16_ablation.txt).  `with_apex` / `use_dali` name dependencies the north star
    drops: asking for them is an error, not a silent fallback."""
    from .cocoeval import COCOeval
    from .data import DataIterator, RotatedDataIterator
    from .utils import Profiler
    if use_dali or with_apex:
        raise RuntimeError('DALI and apex are not part of this build (use the default loader / torch autocast)')
    net = model.module if hasattr(model, 'module') else model
    if not annotations:                                           # every file of the directory, ids by position
        annotations = {'images': [{'id': i, 'file_name': f} for i, f in enumerate(sorted(os.listdir(path)))]}
    if verbose:
        print('Preparing dataset...')
    iterator_class = RotatedDataIterator if rotated_bbox else DataIterator
    # where the model lives decides (validation inside train(): the caller placed it, possibly on the CPU of a GPU host); a
    # stand-alone call on a CPU-resident model moves it to the GPU when there is one, like the reference (infer.py:51-53)
    param_device = next(net.parameters()).device
    if param_device.type == 'cuda' or is_validation or not torch.cuda.is_available():
        device = param_device
    else:
        device = torch.device('cuda', torch.cuda.current_device())
    data_iterator = iterator_class(path, resize, max_size, batch_size, net.stride, max(world, 1), annotations,
                                   training=False, device=device, num_workers=num_workers)
    if verbose:
        print(data_iterator)
    if not is_validation and device.type == 'cuda':
        model = model.to(memory_format=torch.channels_last).cuda()
    was_training = net.training
    model.eval()
    if verbose:
        print('   backend: pytorch')
        print('    device: {} {}'.format(world, 'cpu' if device.type == 'cpu' else 'GPU' if world == 1 else 'GPUs'))
        print('     batch: {}, precision: {}'.format(batch_size, 'mixed' if mixed_precision else 'full'))
        print(' BBOX type:', 'rotated' if rotated_bbox else 'axis aligned')
        print('Running inference...')

    profiler = Profiler(['infer', 'fw'])
    size, batches = len(data_iterator.ids), len(data_iterator)

    def progress(i, forward_seconds):
        profiler.totals['fw'] += forward_seconds
        profiler.counts['fw'] += 1
        profiler.means['fw'] = profiler.totals['fw'] / profiler.counts['fw']
        profiler.bump('infer')
        if verbose and (profiler.totals['infer'] > 60 or i == batches - 1):
            print('[{:{len}}/{}] {:.3f}s/{}-batch (fw: {:.3f}s), {:.1f} im/s'.format(
                min((i + 1) * batch_size, size), size, profiler.means['infer'], batch_size, profiler.means['fw'],
                batch_size / profiler.means['infer'], len=len(str(size))), flush=True)
            profiler.reset()

    has_truth = 'annotations' in data_iterator.coco.dataset
    amp = mixed_precision and device.type == 'cuda'
    # mixed precision = fp16, as in the reference (apex O2 / torch.cuda.amp, infer.py:55-57): 10 mantissa bits; the eager graph
    # under bf16 autocast lost 14 % of the logit amplitude, which is what a model without a fused engine
    # (several backbones, MobileNetV2) would have run through
    with torch.autocast(device.type, dtype=torch.float16, enabled=amp):
        detections = infer_batches(model, data_iterator, rotated_bbox=rotated_bbox,
                                   category_ids=(data_iterator.coco.getCatIds() or None) if has_truth else None,
                                   on_batch=progress)
    if verbose:
        print('Gathering results...')
    if was_training:
        model.train()
    if not is_master:
        return 0
    if not detections:
        print('No detections!')
        return None
    doc = {'annotations': detections, 'images': data_iterator.coco.dataset['images']}
    if 'categories' in data_iterator.coco.dataset:
        doc['categories'] = data_iterator.coco.dataset['categories']
    if detections_file:
        if verbose:
            print('Writing {}...'.format(detections_file))
        for name in ([detections_file] if isinstance(detections_file, str) else detections_file):
            with open(name, 'w') as f:
                json.dump(doc, f, indent=4)
    if has_truth:
        if verbose:
            print('Evaluating model...')
        # reference infer.py:164-168: rotated boxes are scored as regions ('segm' on the corner polygons), not by their
        # axis-aligned fields.  odtk/cocoeval.py computes the polygons' IoU exactly where pycocotools counts mask pixels
        evaluation = COCOeval(data_iterator.coco, data_iterator.coco.loadRes(detections), 'segm' if rotated_bbox else 'bbox')
        if rotated_bbox and verbose:
            print(' (rotated boxes: exact polygon IoU in place of pycocotools\' rasterised masks)')
        evaluation.evaluate()
        evaluation.accumulate()
        evaluation.summarize(out=print if verbose else (lambda line: None))
        return evaluation.stats                                   # mAP and mAR
    return 0
