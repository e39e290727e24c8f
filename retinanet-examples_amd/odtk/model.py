"""RetinaNet (https://arxiv.org/abs/1708.02002) -- model surface of the reference (odtk/model.py)
with the inference post-processing on the MI355X HIP path.

Backbone, FPN, the two 5-conv heads and the losses are stock PyTorch-ROCm modules (MIOpen owns the
MFMA work).  `forward` in eval mode hands the ten head tensors to `odtk.box.detect`, which covers
decode of all five levels + batched NMS in three kernel launches with no host synchronisation
(reference model.py:140-165: python loop of 5 `decode` calls, `torch.cat`, `nms`).

Checkpoint format (`save` / `load`) keeps the reference's keys (model.py:217-258).
"""
import math
import os.path

import numpy as np
import torch
import torch.nn as nn

from . import backbones as backbones_mod
from . import box as box_ops
from .loss import FocalLoss, SmoothL1Loss


class Model(nn.Module):
    def __init__(self, backbones='ResNet50FPN', classes=80, ratios=[1.0, 2.0, 0.5],
                 scales=[4 * 2 ** (i / 3) for i in range(3)], angles=None, rotated_bbox=False,
                 anchor_ious=[0.4, 0.5], config={}):
        super().__init__()
        if not isinstance(backbones, list):
            backbones = [backbones]
        self.backbones = nn.ModuleDict({b: getattr(backbones_mod, b)() for b in backbones})
        self.name = 'RetinaNet'
        self.unused_modules = []
        for b in backbones:
            self.unused_modules.extend(getattr(self.backbones, b).features.unused_modules)
        self.exporting = False
        self.fused_postprocess = True      # False: the reference's op sequence (sigmoid, decode x5, cat, nms)
        self.rotated_bbox = rotated_bbox
        self.anchor_ious = anchor_ious

        self.ratios = ratios
        self.scales = scales
        self.angles = angles if angles is not None else ([-np.pi / 6, 0, np.pi / 6] if rotated_bbox else None)
        self.anchors = {}
        self.classes = classes

        # post-processing hyper-parameters are config entries, not CLI flags (reference model.py:49-52)
        self.threshold = config.get('threshold', 0.05)
        self.top_n = config.get('top_n', 1000)
        self.nms = config.get('nms', 0.5)
        self.detections = config.get('detections', 100)

        self.stride = max(b.stride for b in self.backbones.values())

        def head(out_channels):
            layers = []
            for _ in range(4):
                layers += [nn.Conv2d(256, 256, 3, padding=1), nn.ReLU()]
            layers.append(nn.Conv2d(256, out_channels, 3, padding=1))
            return nn.Sequential(*layers)

        self.num_anchors = len(ratios) * len(scales) * (len(self.angles) if rotated_bbox else 1)
        self.cls_head = head(classes * self.num_anchors)
        self.box_head = head((6 if rotated_bbox else 4) * self.num_anchors)   # rotated: + sin, cos

        self.cls_criterion = FocalLoss()
        self.box_criterion = SmoothL1Loss(beta=0.11)

    def __repr__(self):
        return '\n'.join(['     model: {}'.format(self.name),
                          '  backbone: {}'.format(', '.join(self.backbones.keys())),
                          '   classes: {}, anchors: {}'.format(self.classes, self.num_anchors)])

    def initialize(self, pre_trained=None):
        if pre_trained:
            if not os.path.isfile(pre_trained):
                raise ValueError('No checkpoint {}'.format(pre_trained))
            print('Fine-tuning weights from {}...'.format(os.path.basename(pre_trained)))
            chk = torch.load(pre_trained, map_location='cpu')
            skip = {'cls_head.8.bias', 'cls_head.8.weight'}
            if self.rotated_bbox:
                skip |= {'box_head.8.bias', 'box_head.8.weight'}
            state = self.state_dict()
            state.update({k: v for k, v in chk['state_dict'].items() if k not in skip})
            self.load_state_dict(state)
        else:
            for b in self.backbones.values():
                b.initialize()
            for seq in (self.cls_head, self.box_head):
                for m in seq:
                    if isinstance(m, nn.Conv2d):
                        nn.init.normal_(m.weight, std=0.01)
                        nn.init.zeros_(m.bias)

        # class prior pi = 0.01 on the last classification layer (reference model.py:114-123)
        def prior(layer):
            nn.init.constant_(layer.bias, -math.log((1 - 0.01) / 0.01))
            nn.init.normal_(layer.weight, std=0.01)

        prior(self.cls_head[-1])
        if self.rotated_bbox:
            prior(self.box_head[-1])

    def level_anchors(self, stride):
        if stride not in self.anchors:
            if self.rotated_bbox:
                self.anchors[stride] = box_ops.generate_anchors_rotated(stride, self.ratios, self.scales, self.angles)
            else:
                self.anchors[stride] = box_ops.generate_anchors(stride, self.ratios, self.scales)
        return self.anchors[stride]

    def heads(self, x):
        feats = []
        for b in self.backbones.values():
            feats.extend(b(x))
        return [self.cls_head(t) for t in feats], [self.box_head(t) for t in feats]

    def forward(self, x, rotated_bbox=None):
        if self.training:
            x, targets = x
        cls_heads, box_heads = self.heads(x)
        if self.training:
            return self._compute_loss(x, cls_heads, box_heads, targets.float())

        strides = [x.shape[-1] // c.shape[-1] for c in cls_heads]
        if self.exporting:
            self.strides = strides
            return [c.sigmoid() for c in cls_heads], box_heads
        for s in strides:
            self.level_anchors(s)

        if self.fused_postprocess:
            # sigmoid + decode x5 + nms on the raw head tensors (bf16/fp16/fp32, NCHW or
            # channels_last) in three launches: no sigmoid pass, no .contiguous(), no .float()
            return box_ops.detect(cls_heads, box_heads, strides, self.anchors, self.threshold, self.top_n,
                                  self.nms, self.detections, self.rotated_bbox, logits=True)

        # the reference's sequence, call for call (model.py:140, :153-165)
        cls_heads = [c.sigmoid() for c in cls_heads]
        nms_fn = box_ops.nms_rotated if self.rotated_bbox else box_ops.nms
        decoded = [box_ops.decode(c.contiguous(), b.contiguous(), s, self.threshold, self.top_n, self.anchors[s],
                                  self.rotated_bbox) for c, b, s in zip(cls_heads, box_heads, strides)]
        decoded = [torch.cat(t, 1) for t in zip(*decoded)]
        return nms_fn(*decoded, self.nms, self.detections)

    def _extract_targets(self, targets, stride, size):
        snap = box_ops.snap_to_anchors_rotated if self.rotated_bbox else box_ops.snap_to_anchors
        anchors = self.level_anchors(stride)
        if not self.rotated_bbox:
            anchors = anchors.to(targets.device)
        per_image = [snap(t[t[:, -1] > -1], [s * stride for s in size[::-1]], stride, anchors, self.classes,
                          targets.device, self.anchor_ious) for t in targets]
        return tuple(torch.stack(t) for t in zip(*per_image))

    def _compute_loss(self, x, cls_heads, box_heads, targets):
        cls_sum, box_sum, n_fg = [], [], []
        for cls_head, box_head in zip(cls_heads, box_heads):
            size = cls_head.shape[-2:]
            stride = x.shape[-1] / cls_head.shape[-1]
            cls_target, box_target, depth = self._extract_targets(targets, stride, size)
            n_fg.append((depth > 0).sum().float().clamp(min=1))
            cls_loss = self.cls_criterion(cls_head.view_as(cls_target).float(), cls_target)
            cls_sum.append(((depth >= 0).expand_as(cls_target).float() * cls_loss).sum())
            box_loss = self.box_criterion(box_head.view_as(box_target).float(), box_target)
            box_sum.append(((depth > 0).expand_as(box_target).float() * box_loss).sum())
        n_fg = torch.stack(n_fg).sum()
        return torch.stack(cls_sum).sum() / n_fg, torch.stack(box_sum).sum() / n_fg

    def freeze_unused_params(self):
        for n, p in self.named_parameters():
            if any(u in n for u in self.unused_modules):
                p.requires_grad = False

    def save(self, state):
        checkpoint = {'backbone': list(self.backbones.keys()), 'classes': self.classes,
                      'state_dict': self.state_dict(), 'ratios': self.ratios, 'scales': self.scales}
        if self.rotated_bbox and self.angles:
            checkpoint['angles'] = self.angles
        for key in ('iteration', 'optimizer', 'scheduler'):
            if key in state:
                checkpoint[key] = state[key]
        torch.save(checkpoint, state['path'])

    @classmethod
    def load(cls, filename, rotated_bbox=False):
        if not os.path.isfile(filename):
            raise ValueError('No checkpoint {}'.format(filename))
        checkpoint = torch.load(filename, map_location='cpu')
        kwargs = {k: checkpoint[k] for k in ('ratios', 'scales', 'angles') if k in checkpoint}
        if 'angles' in checkpoint or rotated_bbox:
            kwargs['rotated_bbox'] = True
        model = cls(backbones=checkpoint['backbone'], classes=checkpoint['classes'], **kwargs)
        model.load_state_dict(checkpoint['state_dict'])
        state = {k: checkpoint[k] for k in ('iteration', 'optimizer', 'scheduler') if k in checkpoint}
        return model, state

    def export(self, *args, **kwargs):
        raise NotImplementedError('TensorRT export is not available on MI355X (DALI/TensorRT/DeepStream paths are '
                                  'dropped, BASELINE.json north_star)')
