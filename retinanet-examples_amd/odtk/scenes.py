"""A LEARNABLE synthetic detection set: textured, coloured rectangles of K classes on a noisy background, sizes spanning the
anchor ranges of P3..P6 -- the stand-in for COCO that this image allows (no dataset, no weights, no network).

Why it exists.  The north star asks for "COCO mAP within +-0.1 of the reference" (reference README.md:33:
0.358 for ResNet50FPN; measured by odtk/infer.py:160-172 with pycocotools' COCOeval).  Neither COCO nor a checkpoint is
available here, and `train.SyntheticBatches` -- random boxes on `randn` images -- teaches a network nothing.  A detector
TRAINED on this set by the product's own loop (odtk/train.py) has what a COCO model has and a random-init network has not:
well separated objects, scores spread over (0, 1), boxes regressed to real edges, hundreds of overlapping candidates per
object for the NMS to suppress.  AP on held-out scenes against the TRUE boxes, through the reference pipeline (fp32 eager
graph + the CPU oracle's decode / nms) and through every engine, is then a statement about the model path that random
features cannot make (tools/trained_ap.py -> profiles/r06_trained_ap.txt; tests/test_gpu_trained_ap.py).

A scene is a pure function of (seed, index): the parameters come from a CPU generator seeded with both, the pixels are
rendered on whatever device is asked for with the same float32 arithmetic.  Targets use the reference's format
(odtk/data.py:154-161): per image an [N, 5] table of x, y, w, h, class, padded with -1 rows.
"""
import math

import torch

# class -> (r, g, b) of its fill and the texture laid over it: the colour alone separates the classes for large objects, the
# texture is what P3-sized objects offer once the colour is blurred by the stem
CLASS_COLOURS = [(1.6, -0.8, -0.8), (-0.8, 1.6, -0.8), (-0.8, -0.8, 1.6), (1.4, 1.4, -1.0), (1.4, -1.0, 1.4), (-1.0, 1.4, 1.4),
                 (1.2, 0.2, -1.2), (-1.2, 0.2, 1.2)]
MAX_CLASSES = len(CLASS_COLOURS)


def _textures(height, width, device):
    """[MAX_CLASSES, H, W] multipliers in [0.35, 1]: stripes / checks of class-specific orientation and period."""
    yy = torch.arange(height, device=device, dtype=torch.float32).view(-1, 1).expand(height, width)
    xx = torch.arange(width, device=device, dtype=torch.float32).view(1, -1).expand(height, width)

    def wave(t, period):
        return 0.675 + 0.325 * torch.sign(torch.sin(t * (2 * math.pi / period)))

    return torch.stack([wave(yy, 8), wave(xx, 8), wave(xx, 12) * wave(yy, 12) / 1.0, wave(xx + yy, 11), wave(xx - yy, 11),
                        wave(xx, 16) * wave(yy, 6), wave(yy, 14), wave(xx, 14)])


class SceneBatches:
    """Iterator of (images [B, 3, H, W] float32, targets [B, max_objects, 5]) batches; `rank` / `world` shard a global batch the
    way train.SyntheticBatches does.  `start`: index of the first scene (held-out sets use a range the training never sees)."""

    def __init__(self, batch, height, width, classes=6, max_objects=6, seed=0, rank=0, world=1, device='cpu', length=1 << 30,
                 start=0, min_size=24, max_size=None, noise=0.45):
        if batch % world:
            raise RuntimeError('Batch size should be a multiple of the number of GPUs')
        if not 1 <= classes <= MAX_CLASSES:
            raise ValueError('1..%d classes' % MAX_CLASSES)
        self.batch, self.per_rank, self.rank, self.world = batch, batch // world, rank, world
        self.h, self.w, self.classes, self.max_objects = height, width, classes, max_objects
        self.seed, self.device, self.length, self.start = seed, torch.device(device), length, start
        self.min_size, self.max_size, self.noise = float(min_size), float(max_size or 0.6 * min(height, width)), noise
        self._tex = None

    def __len__(self):
        return self.length

    def __iter__(self):
        for step in range(self.length):
            yield self.batch_at(step)

    # ---- parameters: CPU, a function of (seed, scene index) only -------------------------------------------------------------
    def scene_boxes(self, index):
        """[n, 5] x, y, w, h, class of scene `index` (n = 1..max_objects; sizes log-uniform in [min_size, max_size], aspect
        ratios in [1/2, 2]; a box that overlaps an earlier one with IoU > 0.15 is drawn again: objects may touch and occlude
        each other a little, like real ones, without making the ground truth ambiguous)."""
        g = torch.Generator().manual_seed((self.seed * 1000003 + index) * 2 + 1)
        n = int(torch.randint(1, self.max_objects + 1, (1,), generator=g))
        boxes = []
        for _ in range(n):
            for _attempt in range(20):
                size = math.exp(float(torch.rand(1, generator=g)) * math.log(self.max_size / self.min_size)) * self.min_size
                ratio = math.exp((float(torch.rand(1, generator=g)) - 0.5) * 2 * math.log(2.0))
                w = min(max(round(size * math.sqrt(ratio)), 8), self.w - 2)
                h = min(max(round(size / math.sqrt(ratio)), 8), self.h - 2)
                x = int(torch.randint(0, self.w - w, (1,), generator=g))
                y = int(torch.randint(0, self.h - h, (1,), generator=g))
                c = int(torch.randint(0, self.classes, (1,), generator=g))
                ok = True
                for bx, by, bw, bh, _ in boxes:
                    iw = min(x + w, bx + bw) - max(x, bx)
                    ih = min(y + h, by + bh) - max(y, by)
                    if iw > 0 and ih > 0 and iw * ih / (w * h + bw * bh - iw * ih) > 0.15:
                        ok = False
                        break
                if ok:
                    boxes.append((x, y, w, h, c))
                    break
        return torch.tensor(boxes, dtype=torch.float32).view(-1, 5)

    # ---- pixels ----------------------------------------------------------------------------------------------------------------
    def render(self, indices):
        """Scenes `indices` -> (images [len, 3, H, W] on self.device, targets [len, max_objects, 5])."""
        dev, h, w = self.device, self.h, self.w
        if self._tex is None or self._tex.device != dev:
            self._tex = _textures(h, w, dev)
            self._colours = torch.tensor(CLASS_COLOURS, device=dev)
            self._yy = torch.arange(h, device=dev).view(1, h, 1)
            self._xx = torch.arange(w, device=dev).view(1, 1, w)
        n = len(indices)
        targets = torch.full((n, self.max_objects, 5), -1.0)
        for i, index in enumerate(indices):
            boxes = self.scene_boxes(index)
            targets[i, :boxes.shape[0]] = boxes
        # background: coarse blobs + pixel noise, from a per-batch device generator seeded by the first index (pixels need not
        # be identical across devices, only the boxes)
        g = torch.Generator(device=dev).manual_seed(self.seed * 7919 + int(indices[0]))
        coarse = torch.randn(n, 3, h // 32 + 1, w // 32 + 1, device=dev, generator=g) * 0.5
        images = torch.nn.functional.interpolate(coarse, size=(h, w), mode='bilinear', align_corners=False)
        images = images + torch.randn(n, 3, h, w, device=dev, generator=g) * self.noise
        t = targets.to(dev)
        for k in range(self.max_objects):                                # back to front: later objects occlude earlier ones
            x0, y0, bw, bh, cls = (t[:, k, j].view(n, 1, 1) for j in range(5))
            inside = (cls >= 0) & (self._xx >= x0) & (self._xx < x0 + bw) & (self._yy >= y0) & (self._yy < y0 + bh)   # [n, H, W]
            c = t[:, k, 4].clamp(min=0).long()
            fill = self._colours[c].view(n, 3, 1, 1) * self._tex[c].view(n, 1, h, w)
            images = torch.where(inside.view(n, 1, h, w), fill + 0.15 * images, images)
        return images, targets

    def batch_at(self, step):
        first = self.start + step * self.batch + self.rank * self.per_rank
        images, targets = self.render(list(range(first, first + self.per_rank)))
        return images, targets.to(self.device)


def coco_ground_truth(targets, first_id=0, classes=6):
    """[N, max_objects, 5] targets of scenes first_id, first_id + 1, ... -> the COCO annotation document odtk/cocoeval.py (and
    pycocotools) evaluate against: category ids = class indices, image ids = scene indices."""
    images, annotations = [], []
    for i, rows in enumerate(targets.tolist()):
        images.append({'id': first_id + i, 'file_name': 'scene_%06d' % (first_id + i)})
        for x, y, w, h, c in rows:
            if c < 0:
                continue
            annotations.append({'id': len(annotations) + 1, 'image_id': first_id + i, 'category_id': int(c), 'bbox': [x, y, w, h],
                                'area': w * h, 'iscrowd': 0})
    return {'images': images, 'annotations': annotations, 'categories': [{'id': k, 'name': 'class_%d' % k} for k in range(classes)]}
