"""Training losses of RetinaNet on PyTorch-ROCm: the focal classification loss
(https://arxiv.org/abs/1708.02002) and the smooth-L1 box loss.  Both are elementwise (no reduction);
`Model._compute_loss` masks and sums them.  Values follow the reference's definitions
(reference odtk/loss.py:13-31: alpha = 0.25, gamma = 2, beta = 0.11).

`fused_level_loss` is the MI355X form of one level's share of `Model._compute_loss`
(reference model.py:193-209): focal loss, smooth-L1, both masks and the three sums in ONE hand-written HIP
pass over the head tensors as the convolutions wrote them, and ONE pass in backward (csrc/loss.hpp) -- an
`autograd.Function`, so it drops into the training graph; the torch modules below stay as the reference
implementation it is tested against (tests/test_gpu_loss.py) and as the CPU path."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


def focal_loss(logits, target, alpha=0.25, gamma=2.0):
    """alpha_t * (1 - p_t)^gamma * BCE(logits, target), per element; target is a 0/1 map."""
    prob = torch.sigmoid(logits)
    positive = target == 1
    p_t = torch.where(positive, prob, 1 - prob)                      # probability of the true label
    alpha_t = alpha * target + (1.0 - alpha) * (1.0 - target)
    bce = F.binary_cross_entropy_with_logits(logits, target, reduction='none')
    return alpha_t * (1.0 - p_t) ** gamma * bce


def smooth_l1_loss(pred, target, beta=0.11):
    """|d| - beta/2 beyond beta, d^2 / (2 beta) inside; per element."""
    dist = torch.abs(pred - target)
    quadratic = 0.5 * dist ** 2 / beta
    linear = dist - 0.5 * beta
    return torch.where(dist >= beta, linear, quadratic)


class FocalLoss(nn.Module):
    def __init__(self, alpha=0.25, gamma=2):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    def forward(self, pred_logits, target):
        return focal_loss(pred_logits, target, self.alpha, self.gamma)


class SmoothL1Loss(nn.Module):
    def __init__(self, beta=0.11):
        super().__init__()
        self.beta = beta

    def forward(self, pred, target):
        return smooth_l1_loss(pred, target, self.beta)


class _FusedLevelLoss(torch.autograd.Function):
    """(cls_head, box_head) -> (cls_sum, box_sum, foreground) for one pyramid level of the batch."""

    @staticmethod
    def forward(ctx, cls_head, box_head, depth, box_target, alpha, gamma, beta):
        from . import _C
        sums = _C.retina_loss_forward(cls_head, box_head, depth, box_target, alpha, gamma, beta)
        ctx.save_for_backward(cls_head, box_head, depth, box_target)
        ctx.hyper = (alpha, gamma, beta)
        out = sums.float()
        foreground = out[2]
        ctx.mark_non_differentiable(foreground)
        return out[0], out[1], foreground

    @staticmethod
    def backward(ctx, grad_cls_sum, grad_box_sum, _grad_foreground):
        from . import _C
        cls_head, box_head, depth, box_target = ctx.saved_tensors
        dcls, dbox = _C.retina_loss_backward(cls_head, box_head, depth, box_target, *ctx.hyper, grad_cls_sum, grad_box_sum)
        return dcls, dbox, None, None, None, None, None


def _as_written_pair(cls_head, box_head):
    """Head tensors go to the kernels as the convolutions wrote them; anything exotic is normalised once."""
    pair = [cls_head, box_head]
    for i, t in enumerate(pair):
        if not (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)):
            pair[i] = t.contiguous()
    if pair[0].is_contiguous() != pair[1].is_contiguous() and pair[0].shape[2] * pair[0].shape[3] > 1:
        pair[1] = pair[1].contiguous() if pair[0].is_contiguous() else pair[1].contiguous(memory_format=torch.channels_last)
    if pair[1].dtype != pair[0].dtype:
        pair[1] = pair[1].to(pair[0].dtype)
    return pair


class _FusedPyramidLoss(torch.autograd.Function):
    """(cls_heads..., box_heads...) of ALL levels -> (cls_sums [L], box_sums [L], foreground [L]); one launch each way."""

    @staticmethod
    def forward(ctx, n, alpha, gamma, beta, *tensors):
        from . import _C
        cls_heads, box_heads = tensors[:n], tensors[n:2 * n]
        depths, box_targets = tensors[2 * n:3 * n], tensors[3 * n:4 * n]
        # Through the workspace (per-workgroup sums, added up in a fixed order by a second, tiny launch; csrc/loss.hpp:
        # loss_reduce_kernel): no double atomics at the end of every workgroup -- the walk takes 25-27 us where the atomics
        # form takes 31-35, the reduce launch 4 (profiles/r06_loss_layout_probe.txt) -- and the loss is the same bits on every
        # run.  ODTK_LOSS_ATOMICS=1: the one-launch form of rounds 2-5.
        sums = _C.retina_loss_levels_forward(cls_heads, box_heads, depths, box_targets, alpha, gamma, beta,
                                             reproducible=os.environ.get('ODTK_LOSS_ATOMICS', '0') != '1'
                                             or torch.are_deterministic_algorithms_enabled()).float()
        ctx.save_for_backward(*tensors)
        ctx.meta = (n, alpha, gamma, beta)
        cls_sums, box_sums, foreground = sums[:, 0].contiguous(), sums[:, 1].contiguous(), sums[:, 2].contiguous()
        ctx.mark_non_differentiable(foreground)
        return cls_sums, box_sums, foreground

    @staticmethod
    def backward(ctx, grad_cls_sums, grad_box_sums, _grad_foreground):
        from . import _C
        n, alpha, gamma, beta = ctx.meta
        t = ctx.saved_tensors
        dcls, dbox = _C.retina_loss_levels_backward(t[:n], t[n:2 * n], t[2 * n:3 * n], t[3 * n:4 * n], alpha, gamma, beta,
                                                    grad_cls_sums, grad_box_sums)
        return (None, None, None, None) + tuple(dcls) + tuple(dbox) + (None,) * (2 * n)


def fused_pyramid_loss(cls_heads, box_heads, depths, box_targets, alpha=0.25, gamma=2.0, beta=0.11):
    """`fused_level_loss` for every pyramid level at once: per-level (cls_sums [L], box_sums [L], foreground [L]) from ONE
    HIP launch forward and ONE backward (a training step: two launches where the per-level form needs ten)."""
    from . import _C
    if len(cls_heads) > _C.MAX_LEVELS:              # several backbones: 5 levels each; one launch covers MAX_LEVELS
        groups = [fused_pyramid_loss(cls_heads[i:i + _C.MAX_LEVELS], box_heads[i:i + _C.MAX_LEVELS], depths[i:i + _C.MAX_LEVELS],
                                     box_targets[i:i + _C.MAX_LEVELS], alpha, gamma, beta)
                  for i in range(0, len(cls_heads), _C.MAX_LEVELS)]
        return tuple(torch.cat(parts) for parts in zip(*groups))
    pairs = [_as_written_pair(c, b) for c, b in zip(cls_heads, box_heads)]
    n = len(pairs)
    return _FusedPyramidLoss.apply(n, alpha, gamma, beta, *[p[0] for p in pairs], *[p[1] for p in pairs],
                                   *[d.contiguous() for d in depths], *[t.contiguous() for t in box_targets])


def fused_level_loss(cls_head, box_head, depth, box_target, alpha=0.25, gamma=2.0, beta=0.11):
    """One level's (sum of masked focal losses, sum of masked smooth-L1 losses, number of foreground anchors).

    cls_head [B, A*C, H, W] logits and box_head [B, A*nb, H, W] as the convolutions wrote them (float32 /
    bfloat16 / float16, NCHW or channels_last -- no .float(), no .contiguous()); depth [B, A, 1, H, W] and
    box_target [B, A, nb, H, W] float32 from target assignment.  Equals, and differentiates like,

        cls = FocalLoss(alpha, gamma)(cls_head.view(B, A, C, H, W).float(), onehot(depth - 1))
        (cls * (depth >= 0)).sum(), (SmoothL1Loss(beta)(box_head.view_as(box_target).float(), box_target)
                                     * (depth > 0)).sum(), (depth > 0).sum()"""
    pair = _as_written_pair(cls_head, box_head)
    return _FusedLevelLoss.apply(pair[0], pair[1], depth.contiguous(), box_target.contiguous(), alpha, gamma, beta)
