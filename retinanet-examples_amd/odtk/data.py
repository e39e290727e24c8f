"""Images and COCO-style annotations in, device-resident batches out: the data format on the INPUT side of
the path (reference odtk/data.py; its DALI twin dali.py is dropped by the north star).

Same dataset semantics as the reference -- resize so that the short side is `resize` unless the long side
would exceed `max_size` (data.py:56-59), PIL bilinear resampling, ImageNet mean / std, zero padding (in
normalised space) up to a multiple of the stride and to the largest image of the batch, targets as
`[x, y, w, h(, theta), class]` rows padded with -1, ids + resize ratios for inference -- with a different
split of the work between the host and the GPU:

  * workers hand over **uint8** pixels.  A batch crosses PCIe as ONE `[B, H, W, 4]` uint8 tensor
    (R, G, B, valid): 4 bytes per pixel instead of the reference's 12 (three fp32 planes, normalised by the
    dataset workers one channel at a time, data.py:111-117).
  * normalisation happens on the device, as a 3 x 256-entry table lookup.  The table holds
    `((v / 255) - mean) / std` evaluated in float32 on the host in the reference's operation order, so the
    pixels are bit-identical to the reference's whatever the device's division or fusion rules are; pad
    pixels (valid == 0) become +0.0 exactly like `F.pad` of the normalised image.
  * the batch is born NHWC: `[B, H, W, 3]` viewed as `[B, 3, H, W]` IS a channels_last tensor, the layout
    the convolutions want (the reference builds NCHW and converts every batch, infer.py:75, train.py:96).

`CocoIndex` is the part of `pycocotools.coco.COCO` (nvidia/cocoapi master, un-pinned and absent from this
image) that the reference touches: `dataset`, `imgs`, `getCatIds`, `getAnnIds`, `loadAnns`, `loadImgs`,
`loadRes`.
"""
import json
import math
import os
import random

import numpy as np
import torch
from PIL import Image, ImageEnhance
from torch.utils import data

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


class CocoIndex:
    """Annotation file -> lookup tables, in file order (pycocotools coco.py `createIndex`)."""

    def __init__(self, annotation_file=None, dataset=None):
        if dataset is None:
            if annotation_file is None:
                dataset = {}
            else:
                with open(annotation_file) as f:
                    dataset = json.load(f)
        if not isinstance(dataset, dict):
            raise TypeError('annotation file format {} not supported'.format(type(dataset)))
        self.dataset = dataset
        self.imgs = {im['id']: im for im in dataset.get('images', [])}
        self.cats = {c['id']: c for c in dataset.get('categories', [])}
        self.anns, self.imgToAnns = {}, {}
        for ann in dataset.get('annotations', []):
            self.anns[ann['id']] = ann
            self.imgToAnns.setdefault(ann['image_id'], []).append(ann)

    def getCatIds(self):
        """Category ids in FILE order (coco.py getCatIds with no filter) -- class index k of the network is
        the k-th entry, for training targets and for the detections' `category_id` alike."""
        return [c['id'] for c in self.dataset.get('categories', [])]

    def getImgIds(self):
        return list(self.imgs.keys())

    def getAnnIds(self, imgIds=()):
        ids = [imgIds] if not isinstance(imgIds, (list, tuple)) else imgIds
        if not ids:
            return [a['id'] for a in self.dataset.get('annotations', [])]
        return [a['id'] for i in ids for a in self.imgToAnns.get(i, [])]

    def loadAnns(self, ids=()):
        return [self.anns[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.anns[ids]]

    def loadImgs(self, ids=()):
        return [self.imgs[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.imgs[ids]]

    def loadRes(self, detections):
        """A result set over the same images (coco.py loadRes, bbox branch): every detection gets
        `area = w * h`, `id = position + 1`, `iscrowd = 0`.  Rotated detections keep their polygon but
        are indexed by their axis-aligned extent (mask rasterisation is not part of this port)."""
        known = set(self.imgs)
        anns = []
        for k, det in enumerate(detections):
            if det['image_id'] not in known:
                raise AssertionError('Results do not correspond to current coco set')
            ann = dict(det)
            ann['area'] = det['bbox'][2] * det['bbox'][3]
            ann['id'] = k + 1
            ann['iscrowd'] = 0
            anns.append(ann)
        out = {'images': list(self.dataset.get('images', [])), 'annotations': anns}
        if 'categories' in self.dataset:
            out['categories'] = self.dataset['categories']
        return CocoIndex(dataset=out)


def normalisation_table(dtype=torch.float32):
    """[3, 256]: `((v / 255) - mean) / std` in float32, the reference's operation order (data.py:112-117)."""
    v = torch.arange(256, dtype=torch.float32).div(255)
    rows = [v.clone().sub_(m).div_(s) for m, s in zip(MEAN, STD)]
    return torch.stack(rows).to(dtype)


def normalise_batch(packed, table=None, dtype=torch.float32):
    """`[B, H, W, 4]` uint8 (R, G, B, valid) on any device -> `[B, 3, H, W]` `dtype`, channels_last storage."""
    if table is None:
        table = normalisation_table(dtype)
    table = table.to(device=packed.device, dtype=dtype).reshape(-1)
    index = packed[..., :3].to(torch.int32) + torch.tensor([0, 256, 512], dtype=torch.int32, device=packed.device)
    pixels = torch.where(packed[..., 3:4] != 0, table[index], torch.zeros((), dtype=dtype, device=packed.device))
    return pixels.permute(0, 3, 1, 2)


def _batch_buffer(shape):
    """Uninitialised uint8 tensor for a collated batch.  Inside a loader worker it is born in shared memory, which is
    how the batch reaches the main process anyway (torch's default collate does the same): one first touch of the pages
    instead of two (private buffer, then the copy into a fresh shared segment)."""
    if data.get_worker_info() is None:
        return torch.empty(shape, dtype=torch.uint8)
    proto = torch.empty(0, dtype=torch.uint8)
    storage = proto._typed_storage()._new_shared(math.prod(shape), device=proto.device)
    return proto.new(storage).resize_(*shape)


def _adjust_hue(im, factor):
    """Shift the hue channel by `factor` turns (|factor| <= 0.5), wrapping: what torchvision's PIL
    `adjust_hue` does (the reference calls it, data.py:101-105; torchvision is absent here)."""
    if im.mode in ('L', '1', 'I', 'F'):
        return im
    h, s, v = im.convert('HSV').split()
    shifted = np.array(h, dtype=np.uint8)
    with np.errstate(over='ignore'):
        shifted += np.uint8(int(factor * 255) % 256)
    return Image.merge('HSV', (Image.fromarray(shifted, 'L'), s, v)).convert(im.mode)


class CocoDataset(data.dataset.Dataset):
    """One image (and its boxes) per item.  Items are `(pixels uint8 [h, w, 4] = R, G, B, 255, ...)`: normalisation and
    padding are done per BATCH (see the module docstring); everything else follows reference data.py:13-181.

    The random decisions of training are drawn from `random` in the reference's order (resize jitter,
    quarter-turn, flip, brightness, contrast, hue, saturation), so a seeded run makes the same choices."""

    box_fields = 4

    def __init__(self, path, resize, max_size, stride, annotations=None, training=False, rotate_augment=False,
                 augment_brightness=0.0, augment_contrast=0.0, augment_hue=0.0, augment_saturation=0.0):
        super().__init__()
        self.path = os.path.expanduser(path)
        self.resize, self.max_size, self.stride = resize, max_size, stride
        self.mean, self.std = list(MEAN), list(STD)
        self.training = training
        self.rotate_augment = rotate_augment
        self.augment_brightness, self.augment_contrast = augment_brightness, augment_contrast
        self.augment_hue, self.augment_saturation = augment_hue, augment_saturation
        self.coco = CocoIndex(dataset=annotations) if isinstance(annotations, dict) else CocoIndex(annotations)
        self.ids = list(self.coco.imgs.keys())
        if 'categories' in self.coco.dataset:
            self.categories_inv = {k: i for i, k in enumerate(self.coco.getCatIds())}

    def __len__(self):
        return len(self.ids)

    # -- geometry ---------------------------------------------------------------------------------------
    def _open_resized(self, image_id):
        name = self.coco.loadImgs(image_id)[0]['file_name']
        im = Image.open(os.path.join(self.path, name)).convert('RGB')
        resize = self.resize
        if isinstance(resize, (list, tuple)):
            resize = random.randint(resize[0], resize[-1])
        ratio = resize / min(im.size)
        if ratio * max(im.size) > self.max_size:
            ratio = self.max_size / max(im.size)
        return im.resize(tuple(int(ratio * d) for d in im.size), Image.BILINEAR), ratio

    def _quarter_turn(self, im, boxes, angle):
        """Rotate the image by `angle` in {90, 180, 270} on its own canvas and move the boxes with it
        (reference data.py:68-85)."""
        im = im.rotate(angle)
        x, y, w, h = (boxes[:, k].clone() for k in range(4))
        width, height = im.size
        if angle == 90:
            boxes[:, 0] = y - height / 2 + width / 2
            boxes[:, 1] = width / 2 + height / 2 - x - w
            boxes[:, 2], boxes[:, 3] = h, w
        elif angle == 180:
            boxes[:, 0] = width - x - w
            boxes[:, 1] = height - y - h
        elif angle == 270:
            boxes[:, 0] = width / 2 + height / 2 - y - h
            boxes[:, 1] = x - width / 2 + height / 2
            boxes[:, 2], boxes[:, 3] = h, w
        return im, boxes

    def _flip(self, im, boxes):
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
        boxes[:, 0] = im.size[0] - boxes[:, 0] - boxes[:, 2]
        return im, boxes

    def _colour(self, im):
        if self.augment_brightness:
            im = ImageEnhance.Brightness(im).enhance(max(0, random.normalvariate(1, self.augment_brightness)))
        if self.augment_contrast:
            im = ImageEnhance.Contrast(im).enhance(max(0, random.normalvariate(1, self.augment_contrast)))
        if self.augment_hue:
            im = _adjust_hue(im, min(0.5, max(-0.5, random.normalvariate(0, self.augment_hue))))
        if self.augment_saturation:
            im = ImageEnhance.Color(im).enhance(max(0, random.normalvariate(1, self.augment_saturation)))
        return im

    # -- items --------------------------------------------------------------------------------------------
    def __getitem__(self, index):
        image_id = self.ids[index]
        im, ratio = self._open_resized(image_id)
        target = None
        if self.training:
            boxes, categories = self._get_target(image_id)
            boxes[:, :4] *= ratio
            angle = random.randint(0, 3) * 90
            if self.rotate_augment and angle != 0:
                im, boxes = self._quarter_turn(im, boxes, angle)
            if random.randint(0, 1):
                im, boxes = self._flip(im, boxes)
            im = self._colour(im)
            target = torch.cat([boxes, categories], dim=1)
        # [h, w, 4] = R, G, B, 255: PIL writes the `valid` byte of the batch format, and collate then moves whole rows
        pixels = torch.from_numpy(np.array(im.convert('RGBA'), dtype=np.uint8))
        if self.training:
            return pixels, target
        return pixels, image_id, ratio

    def _get_target(self, image_id):
        """Boxes [N, 4|5] and class indices [N, 1] of one image; a single (1, ..., 1 | -1) row when it has
        none (reference data.py:130-152).  Annotations smaller than a pixel both ways are skipped."""
        boxes, categories = [], []
        for ann in self.coco.loadAnns(self.coco.getAnnIds(imgIds=image_id)):
            if ann['bbox'][2] < 1 and ann['bbox'][3] < 1:
                continue
            boxes.append(self._box_of(ann, image_id))
            cat = ann['category_id']
            if 'categories' in self.coco.dataset:
                cat = self.categories_inv[cat]
            categories.append(cat)
        if boxes:
            return torch.tensor(boxes, dtype=torch.float32), torch.tensor(categories, dtype=torch.float32).unsqueeze(1)
        return torch.ones([1, self.box_fields]), torch.ones([1, 1]) * -1

    def _box_of(self, ann, image_id):
        return list(ann['bbox'])

    # -- batches ------------------------------------------------------------------------------------------
    def collate_fn(self, batch):
        """-> (packed uint8 [B, H, W, 4], targets [B, N, box_fields + 1] padded with -1)   (training)
              (packed uint8 [B, H, W, 4], ids int32 [B], ratios float32 [B, 1, 1])          (inference)
        H, W = the largest image of the batch, each first rounded up to a multiple of the stride
        (reference data.py:119-121, 166-176)."""
        pixels = [item[0] for item in batch]
        stride = self.stride
        up = lambda d: d + (stride - d % stride) % stride
        height = max(up(p.shape[0]) for p in pixels)
        width = max(up(p.shape[1]) for p in pixels)
        packed = _batch_buffer((len(pixels), height, width, 4))
        view = packed.numpy()
        for k, p in enumerate(pixels):
            h, w = p.shape[:2]
            view[k, :h, :w] = p.numpy()                                     # rows of 4 * w contiguous bytes
            view[k, :h, w:] = 0                                             # only the padding is cleared
            view[k, h:] = 0
        if self.training:
            targets = [item[1] for item in batch]
            rows = max(t.shape[0] for t in targets)
            padded = torch.full((len(targets), rows, self.box_fields + 1), -1.0)
            for k, t in enumerate(targets):
                padded[k, :t.shape[0]] = t
            return packed, padded
        ids = torch.tensor([item[1] for item in batch], dtype=torch.int32)
        ratios = torch.tensor([item[2] for item in batch], dtype=torch.float32).view(-1, 1, 1)
        return packed, ids, ratios


class RotatedCocoDataset(CocoDataset):
    """Boxes are `[x, y, w, h, theta]` (theta = 0 appended to plain boxes); quarter turns expand the canvas
    and, with `absolute_angle`, turn theta instead of swapping w and h; a flip negates theta
    (reference data.py:233-415)."""

    box_fields = 5

    def __init__(self, *args, absolute_angle=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.absolute_angle = absolute_angle

    def _box_of(self, ann, image_id):
        box = list(ann['bbox'])
        if len(box) == 4:
            box.append(0.0)
        assert len(box) == 5, 'Bounding box for id %i does not contain five entries.' % image_id
        return box

    def _quarter_turn(self, im, boxes, angle):
        width, height = im.size                                            # before the turn
        im = im.rotate(angle, expand=True)
        x, y, w, h, t = (boxes[:, k].clone() for k in range(5))
        if angle == 90:
            boxes[:, 0], boxes[:, 1] = y, width - x - w
        elif angle == 180:
            boxes[:, 0], boxes[:, 1] = width - x - w, height - y - h
        elif angle == 270:
            boxes[:, 0], boxes[:, 1] = height - y - h, x
        if angle in (90, 270) and not self.absolute_angle:
            boxes[:, 2], boxes[:, 3] = h, w
        if self.absolute_angle:
            t = t + math.radians(angle)
            t = torch.remainder(torch.abs(t), math.pi) * torch.sign(t)
        boxes[:, 4] = t
        return im, boxes

    def _flip(self, im, boxes):
        im, boxes = super()._flip(im, boxes)
        boxes[:, 4] = -boxes[:, 4]
        return im, boxes


class DataIterator:
    """Batches for one rank of a data-parallel job (reference data.py:184-230): `DistributedSampler` shards
    the images when `world > 1`, each rank loads `batch_size // world` of them per step, uploads the uint8
    batch and normalises it on its own GPU.  Yields `(data, targets)` in training and `(data, ids, ratios)`
    in inference, all on `device`; `data` is float `[B, 3, H, W]` with channels_last strides."""

    dataset_class = CocoDataset

    def __init__(self, path, resize, max_size, batch_size, stride, world, annotations, training=False,
                 rotate_augment=False, augment_brightness=0.0, augment_contrast=0.0, augment_hue=0.0,
                 augment_saturation=0.0, device=None, dtype=torch.float32, num_workers=2, rank=None, **dataset_args):
        self.resize, self.max_size = resize, max_size
        self.dataset = self.dataset_class(path, resize=resize, max_size=max_size, stride=stride,
                                          annotations=annotations, training=training, rotate_augment=rotate_augment,
                                          augment_brightness=augment_brightness, augment_contrast=augment_contrast,
                                          augment_hue=augment_hue, augment_saturation=augment_saturation,
                                          **dataset_args)
        self.ids = self.dataset.ids
        self.coco = self.dataset.coco
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
        self.device, self.dtype = torch.device(device), dtype
        self.table = normalisation_table(dtype).to(self.device)
        world = max(1, world)
        if batch_size % world:
            raise RuntimeError('Batch size should be a multiple of the number of GPUs')
        sampler_args = {} if rank is None else {'num_replicas': world, 'rank': rank}
        self.sampler = data.distributed.DistributedSampler(self.dataset, **sampler_args) if world > 1 else None
        self.dataloader = data.DataLoader(self.dataset, batch_size=batch_size // world, sampler=self.sampler,
                                          collate_fn=self.dataset.collate_fn, num_workers=num_workers,
                                          pin_memory=self.device.type == 'cuda',
                                          # training walks the data set epoch after epoch: keep the workers (starting one
                                          # costs ~0.2 s next to an initialised HIP runtime, profiles/r02_loader_probe.txt)
                                          persistent_workers=bool(training and num_workers > 0))

    def __repr__(self):
        return '\n'.join(['    loader: pytorch', '    resize: {}, max: {}'.format(self.resize, self.max_size)])

    def __len__(self):
        return len(self.dataloader)

    def __iter__(self):
        for packed, *rest in self.dataloader:
            images = normalise_batch(packed.to(self.device, non_blocking=True), self.table, self.dtype)
            yield (images, *(t.to(self.device, non_blocking=True) for t in rest))


class RotatedDataIterator(DataIterator):
    """`DataIterator` over `RotatedCocoDataset` (reference data.py:418-484); takes `absolute_angle`."""

    dataset_class = RotatedCocoDataset
