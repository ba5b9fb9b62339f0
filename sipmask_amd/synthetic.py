"""Synthetic workload of BASELINE.json: the reference config dict, reference-init random weights plus
the calibration overrides of SURVEY section 8d (without them random weights give zero detections and a
zero deformable offset, so NMS / mask assembly / the bilinear gather would never run).
"""
import math

import torch

from . import detector  # noqa: F401  (registers SipMask / ResNet / FPN / SipMaskHead / losses)
from .registry import build_detector

# M/configs/sipmask/sipmask_r50_caffe_fpn_gn_1x.py:2-55 (R101: sipmask_r101_caffe_fpn_gn_ms_4x.py)
def model_cfg(depth=50, ssd=False):
    cfg = dict(
        type='SipMask',
        pretrained=None,
        backbone=dict(type='ResNet', depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=False), style='caffe'),
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                  add_extra_convs=True, extra_convs_on_inputs=False, num_outs=5, relu_before_extra_convs=True),
        bbox_head=dict(type='SipMaskHead', num_classes=81, in_channels=256, stacked_convs=4, feat_channels=256,
                       strides=[8, 16, 32, 64, 128],
                       loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                       loss_bbox=dict(type='IoULoss', loss_weight=1.0),
                       loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                       center_sampling=True, center_sample_radius=1.5))
    if ssd:      # M/configs/sipmask/sipmask_r50_caffe_fpn_ssd_6x.py:23-31: two tower convs without GroupNorm, fast_nms
        cfg['bbox_head'].update(stacked_convs=2, ssd_flag=True, norm_cfg=None)
    return cfg


TEST_CFG = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_thr=0.5), max_per_img=100)


SSD_TEST_CFG = dict(TEST_CFG, score_thr=0.1)       # sipmask_r50_caffe_fpn_ssd_6x.py:54-59


def build_synthetic_detector(depth=50, seed=0, bn3_gain=None, ssd=False):
    """Reference init (seeded) + overrides: bn3.weight = bn3_gain (default 1 for R50; 0.5 for R101, whose 33 residual blocks
    at gain 1 push the random-weight activations out of range: every class score NaN, zero detections, and a post-processing
    tail that is cheaper than a real one), non-zero conv_offset, O(1) tower gains, positive box distances, wider
    coefficient spread."""
    if bn3_gain is None:
        bn3_gain = 1.0 if depth <= 50 else 0.5
    torch.manual_seed(seed)
    det = build_detector(model_cfg(depth, ssd), train_cfg=None, test_cfg=dict(SSD_TEST_CFG if ssd else TEST_CFG))
    h = det.bbox_head
    with torch.no_grad():
        for n, p in det.backbone.named_parameters():
            if n.endswith("bn3.weight"):
                p.fill_(bn3_gain)
        torch.nn.init.normal_(h.feat_align.conv_offset.weight, std=0.2)
        for m in list(h.cls_convs) + list(h.reg_convs):
            # (no GroupNorm behind the SSD-style towers: a gain that keeps the std-0.01 init's activations O(1) over two convs)
            m.conv.weight.mul_(6.0 if ssd else 3.0)
        h.feat_align.conv_adaption.weight.mul_(3.0)
        h.fcos_reg.weight.mul_(3.0)
        h.fcos_reg.bias.fill_(2.0)
        h.fcos_cls.weight.mul_(8.0)
        torch.nn.init.normal_(h.sip_cof.weight, std=0.05)
        h.sip_mask_lat.weight.mul_(4.0)
        h.sip_mask_lat0.weight.mul_(4.0)
        for i, s in enumerate(h.scales):
            s.scale.fill_(1.0 + 0.25 * i)
    det.eval()
    return det


def calibrate_cls_bias(det, engine, img, target_per_img=1000, score_thr=0.05):
    """Bisection on fcos_cls.bias so that ~target_per_img class scores per image exceed score_thr.
    Runs the HIP engine once to get the class logits; returns the bias and rebuilds nothing (the
    caller re-prepares the engine afterwards)."""
    engine.run(img)
    torch.cuda.synchronize()
    ncls = engine.ncls
    logits = engine.cls_cof[:, :ncls].float() - float(det.bbox_head.fcos_cls.bias.detach()[0])
    lt = math.log(score_thr / (1 - score_thr))
    lo, hi = -30.0, 30.0
    target = target_per_img * engine.batch
    for _ in range(50):
        mid = 0.5 * (lo + hi)
        if int((logits + mid > lt).sum()) > target:
            hi = mid
        else:
            lo = mid
    b = 0.5 * (lo + hi)
    with torch.no_grad():
        det.bbox_head.fcos_cls.bias.fill_(b)
    return b


def calibrate_offset_scale(det, engine, img, target_std=1.0):
    """Rescale FeatureAlign.conv_offset so that the sampling offsets of the synthetic net have `target_std` pixels (the GN
    configs get ~1 px from the overrides above: 0.3 % of the components beyond 3 px; the SSD-style towers have no norm layer,
    so the same weights give tens of pixels -- every tap far from its position, which no trained FeatureAlign produces).
    Runs the engine once; the caller re-prepares the plan afterwards."""
    engine.run(img)
    torch.cuda.synchronize()
    std = float(engine.offsets.float().std())
    if std > 0:
        with torch.no_grad():
            det.bbox_head.feat_align.conv_offset.weight.mul_(target_std / std)
    return std


# V/configs/sipmask/sipmask_r50_caffe_fpn_gn_1x.py (V/ = SipMask-VIS/): 41 classes, stacked_convs=3, test_cfg :51-56
VIS_TEST_CFG = dict(nms_pre=200, min_bbox_size=0, score_thr=0.03, nms=dict(type='nms', iou_thr=0.5), max_per_img=10)


def build_synthetic_vis_detector(seed=0):
    """BASELINE config #5: SipMask-VIS R50 (track head), reference init + the same calibration overrides."""
    from . import vis_head  # noqa: F401  (registers SipMaskVIS / SipMaskVISHead)
    torch.manual_seed(seed)
    cfg = model_cfg(50)
    cfg['type'] = 'SipMaskVIS'
    cfg['bbox_head'].update(type='SipMaskVISHead', num_classes=41, stacked_convs=3)
    det = build_detector(cfg, train_cfg=None, test_cfg=dict(VIS_TEST_CFG))
    h = det.bbox_head
    with torch.no_grad():
        for n, p in det.backbone.named_parameters():
            if n.endswith("bn3.weight"):
                p.fill_(1.0)
        torch.nn.init.normal_(h.feat_align.conv_offset.weight, std=0.2)
        for m in list(h.cls_convs) + list(h.reg_convs) + list(h.track_convs):
            m.conv.weight.mul_(3.0)
        h.feat_align.conv_adaption.weight.mul_(3.0)
        h.fcos_reg.weight.mul_(3.0)
        h.fcos_reg.bias.fill_(2.0)
        h.fcos_cls.weight.mul_(8.0)
        torch.nn.init.normal_(h.sip_cof.weight, std=0.05)
        h.sip_mask_lat.weight.mul_(4.0)
        h.sip_mask_lat0.weight.mul_(4.0)
    det.eval()
    return det
