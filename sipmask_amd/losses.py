"""Loss modules under the reference registry names (training rows a14/a15).

  FocalLoss         M/mmdet/models/losses/focal_loss.py:45-82 -> HIP sigmoid focal loss
  IoULoss           M/mmdet/models/losses/iou_loss.py:9-27,73-96 (kept in ATen)
  CrossEntropyLoss  M/mmdet/models/losses/cross_entropy_loss.py:35-51,60-103 (sigmoid variant)
  MSELoss           M/mmdet/models/losses/mse_loss.py
Reduction rules: M/mmdet/models/losses/utils.py:6-52.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .ops import sigmoid_focal_loss
from .registry import LOSSES


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        if reduction == 'sum':
            return loss.sum()
        return loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


@LOSSES.register_module
class FocalLoss(nn.Module):

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        loss = sigmoid_focal_loss(pred, target, self.gamma, self.alpha)
        if weight is not None:
            weight = weight.view(-1, 1)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


def _aligned_iou(b1, b2):
    lt = torch.max(b1[:, :2], b2[:, :2])
    rb = torch.min(b1[:, 2:], b2[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    a1 = (b1[:, 2] - b1[:, 0] + 1) * (b1[:, 3] - b1[:, 1] + 1)
    a2 = (b2[:, 2] - b2[:, 0] + 1) * (b2[:, 3] - b2[:, 1] + 1)
    return overlap / (a1 + a2 - overlap)


@LOSSES.register_module
class IoULoss(nn.Module):

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        if weight is not None and not torch.any(weight > 0):
            return (pred * weight).sum()
        reduction = reduction_override if reduction_override else self.reduction
        loss = -_aligned_iou(pred, target).clamp(min=self.eps).log()
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module
class CrossEntropyLoss(nn.Module):

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid and not use_mask, "SipMask uses the sigmoid (BCE-with-logits) variant only"
        self.use_sigmoid, self.reduction, self.loss_weight = use_sigmoid, reduction, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None:
            weight = weight.float()
        loss = F.binary_cross_entropy_with_logits(cls_score, label.float(), reduction='none')
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module
class MSELoss(nn.Module):

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        return self.loss_weight * weight_reduce_loss(F.mse_loss(pred, target, reduction='none'), weight,
                                                     self.reduction, avg_factor)
