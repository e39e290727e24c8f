"""Training losses of RetinaNet on PyTorch-ROCm: the focal classification loss
(https://arxiv.org/abs/1708.02002) and the smooth-L1 box loss.  Both are elementwise (no reduction);
`Model._compute_loss` masks and sums them.  Values follow the reference's definitions
(reference odtk/loss.py:13-31: alpha = 0.25, gamma = 2, beta = 0.11); the north star keeps these off
the hand-written HIP path."""
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
