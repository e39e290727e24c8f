"""RetinaNet (https://arxiv.org/abs/1708.02002) on MI355X.

Public surface = the reference's `odtk.model.Model` (constructor arguments, `initialize`, `forward`
in train / eval mode, `save` / `load` with the same checkpoint keys, reference odtk/model.py), so
training scripts and checkpoints carry over.  What runs where:

  backbone + FPN + the two 5-conv heads + losses   stock PyTorch-ROCm modules (MIOpen owns the MFMA work)
  eval forward on a GPU (the default)               the BN-folded inference engine (odtk/fused.py: folded
                                                    weights, HIP bias/skip/ReLU epilogues, 1x1 convs as
                                                    fused GEMMs) + `odtk.box.detect`: sigmoid + decode of all
                                                    five levels + batched NMS, hand-written HIP, six launches,
                                                    no host sync, head tensors read in place.  Built lazily
                                                    from the current weights, rebuilt when they change.
  `fused_graph = False`                             the eager nn.Module graph (what training uses) + the same
                                                    fused post-processing
  `fused_postprocess = False`                       the reference's own op sequence (model.py:140-165)
                                                    on the same kernels, for A/B checks
  CPU tensors                                       eager graph + the pure-torch decode / nms branch of
                                                    odtk/box.py (BASELINE config 0: no GPU needed)
"""
import math
import os.path

import numpy as np
import torch
import torch.nn as nn

from . import backbones as backbones_mod
from . import box as box_ops
from .loss import FocalLoss, SmoothL1Loss, fused_pyramid_loss

DEFAULT_RATIOS = [1.0, 2.0, 0.5]
DEFAULT_SCALES = [4 * 2 ** (i / 3) for i in range(3)]
DEFAULT_ANGLES = [-np.pi / 6, 0, np.pi / 6]
CLASS_PRIOR = 0.01                       # initial foreground probability of every anchor


def conv_tower(out_channels, width=256, depth=4):
    """`depth` x (3x3 conv + ReLU) followed by the 3x3 output conv; shared by all pyramid levels.
    nn.Sequential indices (0, 2, 4, 6, 8 are convs) are part of the checkpoint format."""
    layers = []
    for _ in range(depth):
        layers.extend([nn.Conv2d(width, width, 3, padding=1), nn.ReLU()])
    layers.append(nn.Conv2d(width, out_channels, 3, padding=1))
    return nn.Sequential(*layers)


def _init_tower(tower):
    for layer in tower:
        if isinstance(layer, nn.Conv2d):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.zeros_(layer.bias)


def _init_prior(conv):
    nn.init.normal_(conv.weight, std=0.01)
    nn.init.constant_(conv.bias, -math.log((1 - CLASS_PRIOR) / CLASS_PRIOR))


class Model(nn.Module):
    CHECKPOINT_EXTRAS = ('iteration', 'optimizer', 'scheduler')

    def __init__(self, backbones='ResNet50FPN', classes=80, ratios=DEFAULT_RATIOS, scales=DEFAULT_SCALES,
                 angles=None, rotated_bbox=False, anchor_ious=[0.4, 0.5], config={}):
        super().__init__()
        names = backbones if isinstance(backbones, list) else [backbones]
        self.backbones = nn.ModuleDict({n: getattr(backbones_mod, n)() for n in names})
        self.name = 'RetinaNet'
        self.unused_modules = [u for n in names for u in self.backbones[n].features.unused_modules]
        self.stride = max(b.stride for b in self.backbones.values())

        self.classes = classes
        self.ratios, self.scales = ratios, scales
        self.rotated_bbox = rotated_bbox
        self.angles = angles if angles is not None else (DEFAULT_ANGLES if rotated_bbox else None)
        self.anchor_ious = anchor_ious
        self.anchors = {}                                   # stride -> base anchors, filled lazily

        # post-processing hyper-parameters are config entries, not CLI flags (reference model.py:49-52)
        self.threshold = config.get('threshold', 0.05)
        self.top_n = config.get('top_n', 1000)
        self.nms = config.get('nms', 0.5)
        self.detections = config.get('detections', 100)
        self.exporting = False
        self.fused_postprocess = True
        self.fused_graph = True                             # eval on a GPU runs the BN-folded engine (odtk/fused.py)
        self.fused_loss = True                              # training on a GPU: HIP focal / smooth-L1 reduction (csrc/loss.hpp)
        self.__dict__['_engine_cache'] = {}                 # (not a submodule: keeps state_dict / checkpoints unchanged)

        self.num_anchors = len(ratios) * len(scales) * (len(self.angles) if rotated_bbox else 1)
        box_params = 6 if rotated_bbox else 4               # rotated: (dx, dy, dw, dh, sin, cos)
        self.cls_head = conv_tower(classes * self.num_anchors)
        self.box_head = conv_tower(box_params * self.num_anchors)
        self.cls_criterion = FocalLoss()
        self.box_criterion = SmoothL1Loss(beta=0.11)

    def __repr__(self):
        return '\n'.join(['     model: {}'.format(self.name),
                          '  backbone: {}'.format(', '.join(self.backbones.keys())),
                          '   classes: {}, anchors: {}'.format(self.classes, self.num_anchors)])

    # ------------------------------------------------------------------ weights
    def initialize(self, pre_trained=None):
        """Fresh initialisation, or fine-tuning from a checkpoint whose last classification (and, for
        rotated boxes, regression) layer is re-initialised (reference model.py:80-123)."""
        if pre_trained:
            if not os.path.isfile(pre_trained):
                raise ValueError('No checkpoint {}'.format(pre_trained))
            print('Fine-tuning weights from {}...'.format(os.path.basename(pre_trained)))
            donor = torch.load(pre_trained, map_location='cpu')['state_dict']
            dropped = ['cls_head.8.'] + (['box_head.8.'] if self.rotated_bbox else [])
            state = self.state_dict()
            state.update({k: v for k, v in donor.items() if not any(k.startswith(d) for d in dropped)})
            self.load_state_dict(state)
        else:
            for backbone in self.backbones.values():
                backbone.initialize()
            _init_tower(self.cls_head)
            _init_tower(self.box_head)
        _init_prior(self.cls_head[-1])
        if self.rotated_bbox:
            _init_prior(self.box_head[-1])

    def freeze_unused_params(self):
        for name, param in self.named_parameters():
            if any(u in name for u in self.unused_modules):
                param.requires_grad = False

    def save(self, state):
        checkpoint = {'backbone': list(self.backbones.keys()), 'classes': self.classes, 'ratios': self.ratios,
                      'scales': self.scales, 'state_dict': self.state_dict()}
        if self.rotated_bbox and self.angles:
            checkpoint['angles'] = self.angles
        checkpoint.update({k: state[k] for k in self.CHECKPOINT_EXTRAS if k in state})
        torch.save(checkpoint, state['path'])

    @classmethod
    def load(cls, filename, rotated_bbox=False):
        if not os.path.isfile(filename):
            raise ValueError('No checkpoint {}'.format(filename))
        checkpoint = torch.load(filename, map_location='cpu')
        kwargs = {k: checkpoint[k] for k in ('ratios', 'scales', 'angles') if k in checkpoint}
        if rotated_bbox or 'angles' in checkpoint:
            kwargs['rotated_bbox'] = True
        model = cls(backbones=checkpoint['backbone'], classes=checkpoint['classes'], **kwargs)
        model.load_state_dict(checkpoint['state_dict'])
        return model, {k: checkpoint[k] for k in cls.CHECKPOINT_EXTRAS if k in checkpoint}

    def export(self, *args, **kwargs):
        raise NotImplementedError('TensorRT export is not available on MI355X (the DALI / TensorRT / DeepStream '
                                  'paths are dropped, BASELINE.json north_star)')

    def fuse(self, dtype=torch.bfloat16):
        """Inference engine with BN folded into the convolutions and the HIP epilogue (odtk/fused.py)."""
        from .fused import FusedRetinaNet
        return FusedRetinaNet(self, dtype)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__['_engine_cache'].clear()              # .to() / .cuda() / .float(): new storages
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode=True):
        self.__dict__['_engine_rescan'] = True              # next eval call re-reads WHICH tensors the model is made of
        return super().train(mode)

    def invalidate_engine(self):
        """Drop the cached inference engine (and its hipGraphs).  Needed after weight changes the cache cannot see: writes
        through `.data` (`p.data.copy_()`, EMA swaps -- a separate version counter) between two eval calls."""
        self.__dict__['_engine_cache'].clear()

    def inference_engine(self, dtype):
        """The cached BN-folded engine for `dtype`; None when this model has no fused form (several
        backbones, exotic blocks).  The engine is a SNAPSHOT of the weights, so it remembers the tensors it
        was folded from together with their version counters and storage addresses, and is rebuilt when
        any of them moved: optimizer steps and load_state_dict write in place (version bump), .to() /
        .cuda() swap storages (cache dropped in `_apply`)."""
        from .fused import FusedRetinaNet
        cache = self.__dict__['_engine_cache']
        hit = cache.get(dtype)
        if hit is not None:
            tensors, stamp, engine = hit
            if self.__dict__.pop('_engine_rescan', False):
                # after every train() / eval() switch: is the model still MADE OF the tensors the engine was folded from?
                # (a replaced submodule or Parameter keeps the old objects alive in `tensors`, version and address unchanged).
                # The module walk costs ~0.4 ms, so it is not repeated on every call; between two switches use invalidate_engine()
                current = list(self.parameters()) + list(self.buffers())
                if len(current) != len(tensors) or any(a is not b for a, b in zip(current, tensors)):
                    hit = None
            if hit is not None and [t._version for t in tensors] + [t.data_ptr() for t in tensors] == stamp:
                return engine
        if not FusedRetinaNet.supports(self):
            return None
        cache.clear()                                       # one engine at a time: it holds a copy of the weights
        tensors = list(self.parameters()) + list(self.buffers())
        engine = FusedRetinaNet(self, dtype)
        cache[dtype] = (tensors, [t._version for t in tensors] + [t.data_ptr() for t in tensors], engine)
        self.__dict__.pop('_engine_rescan', None)
        return engine

    # ------------------------------------------------------------------ forward
    def level_anchors(self, stride):
        if stride not in self.anchors:
            if self.rotated_bbox:
                self.anchors[stride] = box_ops.generate_anchors_rotated(stride, self.ratios, self.scales, self.angles)
            else:
                self.anchors[stride] = box_ops.generate_anchors(stride, self.ratios, self.scales)
        return self.anchors[stride]

    def heads(self, x):
        """Raw head tensors of every pyramid level: (cls logits x5, box deltas x5)."""
        pyramid = [feature for backbone in self.backbones.values() for feature in backbone(x)]
        return [self.cls_head(f) for f in pyramid], [self.box_head(f) for f in pyramid]

    def forward(self, x, rotated_bbox=None, graph=False):
        """Reference surface (model.py:131): training -> (cls_loss, box_loss) of (images, targets); eval -> (scores, boxes,
        classes).  graph=True (eval on a GPU, opt-in): the whole call -- engine + post-processing, ~300 launches -- is captured
        once per input geometry into ONE hipGraph and replayed (batch 1: launch-bound eager); the graph
        belongs to the engine and goes with it when the weights change."""
        if self.training:
            if graph:
                raise RuntimeError('Model.forward(graph=True) is an inference option (eval mode)')
            images, targets = x
            cls_heads, box_heads = self.heads(images)
            return self._compute_loss(images, cls_heads, box_heads, targets.float())

        if self.fused_graph and self.fused_postprocess and x.is_cuda and not self.exporting:
            # the default inference path: same function as the eager graph below up to the rounding of the
            # folded weights (tests/test_gpu_fused_model.py, tests/test_gpu_detection_parity.py)
            dtype = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else self.cls_head[0].weight.dtype
            engine = self.inference_engine(dtype)
            if engine is not None:
                return engine.replay(x) if graph else engine(x)
        if graph:
            raise RuntimeError('Model.forward(graph=True) needs the fused inference engine (eval mode, GPU tensors, a single '
                               'ResNet-FPN backbone)')

        cls_heads, box_heads = self.heads(x)
        strides = [x.shape[-1] // c.shape[-1] for c in cls_heads]
        if self.exporting:
            self.strides = strides
            return [c.sigmoid() for c in cls_heads], box_heads
        for stride in strides:
            self.level_anchors(stride)
        return self.postprocess(cls_heads, box_heads, strides)

    def postprocess(self, cls_heads, box_heads, strides):
        """Raw head tensors -> (scores [B, D], boxes [B, D, 4|6], classes [B, D])."""
        if self.fused_postprocess and cls_heads[0].is_cuda:
            # sigmoid + decode x5 + nms on the head tensors as the convolutions wrote them
            return box_ops.detect(cls_heads, box_heads, strides, self.anchors, self.threshold, self.top_n,
                                  self.nms, self.detections, self.rotated_bbox, logits=True)
        # the reference's sequence, call for call (model.py:140, :153-165); on CPU tensors this is the
        # pure-torch branch of odtk/box.py
        suppress = box_ops.nms_rotated if self.rotated_bbox else box_ops.nms
        per_level = [box_ops.decode(c.sigmoid().contiguous(), b.contiguous(), s, self.threshold, self.top_n,
                                    self.anchors[s], self.rotated_bbox)
                     for c, b, s in zip(cls_heads, box_heads, strides)]
        return suppress(*[torch.cat(parts, 1) for parts in zip(*per_level)], self.nms, self.detections)

    # ------------------------------------------------------------------ training loss
    def _extract_targets(self, targets, stride, size, want_cls_target=True):
        """Per-image target assignment of one level, stacked over the batch (reference model.py:167-184)."""
        anchors = self.level_anchors(stride)
        if targets.is_cuda and not self.rotated_bbox:
            # one fused HIP launch for the whole batch (csrc/targets.hpp) instead of ~25 torch ops per image
            return box_ops.snap_to_anchors_batched(targets, size[1], size[0], stride, anchors, self.classes,
                                                   self.anchor_ious, want_cls_target)
        assign = box_ops.snap_to_anchors_rotated if self.rotated_bbox else box_ops.snap_to_anchors
        if not self.rotated_bbox:
            anchors = anchors.to(targets.device)
        pixels = [extent * stride for extent in size[::-1]]             # [W, H] of the padded image
        per_image = [assign(t[t[:, -1] > -1], pixels, stride, anchors, self.classes, targets.device, self.anchor_ious)
                     for t in targets]
        return tuple(torch.stack(parts) for parts in zip(*per_image))

    def _compute_loss(self, x, cls_heads, box_heads, targets):
        """Focal + smooth-L1 losses summed over levels and normalised by the number of foreground
        anchors (reference model.py:186-210); `depth` is -1 ignore / 0 background / class+1."""
        cls_total, box_total, foreground = 0.0, 0.0, 0.0
        if self.fused_loss and cls_heads[0].is_cuda:
            # targets of every level (ONE HIP launch for all of them; the class map is implied by depth and not even built), then focal
            # + smooth-L1 + masks + sums of ALL levels in one HIP pass -- and one more in backward (csrc/loss.hpp)
            if len(cls_heads) <= box_ops.MAX_LEVELS_PER_CALL:
                strides = [x.shape[-1] / c.shape[-1] for c in cls_heads]
                assign = box_ops.snap_to_anchors_rotated_levels if self.rotated_bbox else box_ops.snap_to_anchors_levels
                _, box_targets, depths = assign(
                    targets, [tuple(c.shape[-2:]) for c in cls_heads], strides, [self.level_anchors(s) for s in strides],
                    self.classes, self.anchor_ious, want_cls_target=False)
            else:
                depths, box_targets = [], []
                for cls_head in cls_heads:
                    stride = x.shape[-1] / cls_head.shape[-1]
                    _, box_target, depth = self._extract_targets(targets, stride, cls_head.shape[-2:], False)
                    depths.append(depth)
                    box_targets.append(box_target)
            cls_sums, box_sums, fg = fused_pyramid_loss(cls_heads, box_heads, depths, box_targets, self.cls_criterion.alpha,
                                                        self.cls_criterion.gamma, self.box_criterion.beta)
            foreground = fg.clamp(min=1).sum()              # per level, as the reference clamps (model.py:196)
            return cls_sums.sum() / foreground, box_sums.sum() / foreground
        for cls_head, box_head in zip(cls_heads, box_heads):
            stride = x.shape[-1] / cls_head.shape[-1]
            cls_target, box_target, depth = self._extract_targets(targets, stride, cls_head.shape[-2:])
            foreground = foreground + (depth > 0).sum().float().clamp(min=1)
            cls_loss = self.cls_criterion(cls_head.view_as(cls_target).float(), cls_target)
            cls_total = cls_total + (cls_loss * (depth >= 0).expand_as(cls_target).float()).sum()
            box_loss = self.box_criterion(box_head.view_as(box_target).float(), box_target)
            box_total = box_total + (box_loss * (depth > 0).expand_as(box_target).float()).sum()
        return cls_total / foreground, box_total / foreground
