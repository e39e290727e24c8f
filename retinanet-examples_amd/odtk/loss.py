"""Focal loss (https://arxiv.org/abs/1708.02002) and smooth-L1, elementwise, on PyTorch-ROCm
(reference odtk/loss.py:5-31; the north star keeps these off the hand-written path)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class FocalLoss(nn.Module):
    def __init__(self, alpha=0.25, gamma=2):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    def forward(self, pred_logits, target):
        p = pred_logits.sigmoid()
        bce = F.binary_cross_entropy_with_logits(pred_logits, target, reduction='none')
        weight = target * self.alpha + (1. - target) * (1. - self.alpha)
        p_true = torch.where(target == 1, p, 1 - p)
        return weight * (1. - p_true) ** self.gamma * bce


class SmoothL1Loss(nn.Module):
    def __init__(self, beta=0.11):
        super().__init__()
        self.beta = beta

    def forward(self, pred, target):
        err = (pred - target).abs()
        return torch.where(err >= self.beta, err - 0.5 * self.beta, 0.5 * err ** 2 / self.beta)
