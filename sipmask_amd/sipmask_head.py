"""SipMaskHead -- drop-in for the reference ``@HEADS`` entry
(M/mmdet/models/anchor_heads/sipmask_head.py:107-960): same constructor kwargs and defaults
(:109-133), same parameter names, same call protocol
(forward / get_bboxes / loss / init_weights).  The arithmetic runs in the HIP engine.
"""
import numpy as np
import torch
import torch.nn as nn

from . import hip_ops as H
from .modules import ConvModule, bias_init_with_prob, normal_init
from .ops import CropSplit, CropSplitGt, DeformConv, Scale
from .registry import HEADS, build_loss

INF = 1e8


class FeatureAlign(nn.Module):
    """sipmask_head.py:21-55: conv_offset (1x1, 4 -> G*18, no bias) + DeformConv 3x3 + GN(32) + ReLU."""

    def __init__(self, in_channels, out_channels, kernel_size=3, deformable_groups=4, flag_norm=True):
        super().__init__()
        offset_channels = kernel_size * kernel_size * 2
        self.conv_offset = nn.Conv2d(4, deformable_groups * offset_channels, 1, bias=False)
        self.conv_adaption = DeformConv(in_channels, out_channels, kernel_size=kernel_size,
                                        padding=(kernel_size - 1) // 2, deformable_groups=deformable_groups)
        self.relu = nn.ReLU(inplace=True)
        self.norm = nn.GroupNorm(32, in_channels)
        self.flag_norm = flag_norm

    def init_weights(self, bias_value=0):
        torch.nn.init.normal_(self.conv_offset.weight, std=0.0)
        torch.nn.init.normal_(self.conv_adaption.weight, std=0.01)


@HEADS.register_module
class SipMaskHead(nn.Module):

    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=4, strides=(4, 8, 16, 32, 64),
                 regress_ranges=((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF)), center_sampling=False,
                 center_sample_radius=1.5, ssd_flag=False, rescoring_flag=False,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox=dict(type='IoULoss', loss_weight=1.0),
                 loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 conv_cfg=None, norm_cfg=dict(type='GN', num_groups=32, requires_grad=True)):
        super().__init__()
        if rescoring_flag:
            raise NotImplementedError("rescoring_flag (SipMask++ mask-IoU branch) is a 'next' row (SURVEY 8f)")
        self.num_classes = num_classes
        self.cls_out_channels = num_classes - 1
        self.in_channels, self.feat_channels, self.stacked_convs = in_channels, feat_channels, stacked_convs
        self.strides, self.regress_ranges = strides, regress_ranges
        from . import losses  # noqa: F401  (registers FocalLoss / IoULoss / CrossEntropyLoss / MSELoss)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.loss_centerness = build_loss(loss_centerness)
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        self.fp16_enabled = False
        self.center_sampling, self.center_sample_radius = center_sampling, center_sample_radius
        self.ssd_flag, self.rescoring_flag = ssd_flag, rescoring_flag
        self.nc = 32
        self._engines = {}
        self._init_layers()

    def _init_layers(self):
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(self.stacked_convs - 1):                       # 3 cls convs (:161)
            chn = self.in_channels if i == 0 else self.feat_channels
            self.cls_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1, conv_cfg=self.conv_cfg,
                                             norm_cfg=self.norm_cfg, bias=self.norm_cfg is None))
        for i in range(self.stacked_convs):                           # 4 reg convs (:174)
            chn = self.in_channels if i == 0 else self.feat_channels
            self.reg_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1, conv_cfg=self.conv_cfg,
                                             norm_cfg=self.norm_cfg, bias=self.norm_cfg is None))
        self.fcos_cls = nn.Conv2d(self.feat_channels, self.cls_out_channels, 3, padding=1)
        self.fcos_reg = nn.Conv2d(self.feat_channels, 4, 3, padding=1)
        self.fcos_centerness = nn.Conv2d(self.feat_channels, 1, 3, padding=1)
        self.scales = nn.ModuleList([Scale(1.0) for _ in self.strides])
        self.feat_align = FeatureAlign(self.feat_channels, self.feat_channels, 3, flag_norm=self.norm_cfg is not None)
        self.sip_cof = nn.Conv2d(self.feat_channels, self.nc * 4, 3, padding=1)
        self.sip_mask_lat = nn.Conv2d(512, self.nc, 3, padding=1)
        self.sip_mask_lat0 = nn.Conv2d(768, 512, 1, padding=0)
        self.relu = nn.ReLU(inplace=True)
        self.crop_cuda = CropSplit(2)
        self.crop_gt_cuda = CropSplitGt(2)
        self.init_weights()

    def init_weights(self):
        for m in self.cls_convs:
            normal_init(m.conv, std=0.01)
        for m in self.reg_convs:
            normal_init(m.conv, std=0.01)
        normal_init(self.fcos_cls, std=0.01, bias=bias_init_with_prob(0.01))
        normal_init(self.fcos_reg, std=0.01)
        normal_init(self.fcos_centerness, std=0.01)
        normal_init(self.sip_cof, std=0.001)
        normal_init(self.sip_mask_lat, std=0.01)
        normal_init(self.sip_mask_lat0, std=0.01)
        self.feat_align.init_weights()
        self._engines = {}

    # ------------------------------------------------------------------ forward (HIP engine, head mode)
    def _engine(self, batch, sizes, img_shape=None, cfg=None):
        from .engine import SipMaskEngine
        key = (batch, tuple(sizes), tuple(img_shape or ()), repr(cfg))
        eng = self._engines.get(key)
        if eng is None:
            sd = {"bbox_head." + k: v for k, v in self.state_dict().items()}
            eng = SipMaskEngine.for_head(sd, batch, sizes, num_classes=self.num_classes, strides=self.strides,
                                         test_cfg=cfg, img_shape=img_shape, ssd_flag=self.ssd_flag)
            self._engines = {key: eng}     # one cached plan; weights are snapshotted at build time
        return eng

    def forward(self, feats):
        """feats: tuple of 5 NCHW float tensors -> (cls_scores, bbox_preds, centernesses, cof_preds, feat_masks)."""
        b = feats[0].shape[0]
        sizes = [tuple(f.shape[-2:]) for f in feats]
        eng = self._engine(b, sizes)
        eng.load_pyramid(feats)
        eng.run_head()
        return eng.head_outputs()

    def get_masks(self, cls_scores, bbox_preds, centernesses, cof_preds, feat_masks, img_metas, cfg, rescale=None):
        """Tensor-only get_bboxes (sipmask_head.py:500-633 up to, not including, RLE): returns per image
        (det_bboxes [N,5], det_labels [N], idxs_keep [N], masks uint8 [N,Ho,Wo])."""
        from .engine import PostProcessor
        post = PostProcessor(cls_scores, bbox_preds, centernesses, cof_preds, feat_masks, img_metas, cfg,
                             self.strides, rescale, self.ssd_flag)
        return post.run()

    def get_bboxes(self, cls_scores, bbox_preds, centernesses, cof_preds, feat_masks, img_metas, cfg, rescale=None):
        """Reference packaging (sipmask_head.py:645-662): list of (det_bboxes, det_labels, cls_segms) with
        cls_segms[label] = list of COCO RLE dicts {'size': [H, W], 'counts': bytes}.  The masks are pasted on the
        ori_shape / img_shape canvas and run-length encoded ON DEVICE (sm_rle_encode), so the per-detection
        `mask.cpu()` + `mask_util.encode` loop of the reference becomes two small D2H copies per batch."""
        from .engine import PostProcessor
        post = PostProcessor(cls_scores, bbox_preds, centernesses, cof_preds, feat_masks, img_metas, cfg,
                             self.strides, rescale, self.ssd_flag)
        res = post.run()
        meta = img_metas[0]
        shp = meta['ori_shape'] if rescale else meta['img_shape']
        rles = post.encode_rle(shp[:2])
        out = []
        for (det, labels, _, _), rle in zip(res, rles):
            cls_segms = [[] for _ in range(self.num_classes - 1)]
            for lab, r in zip(labels.cpu().tolist(), rle):
                cls_segms[lab].append(r)
            out.append((det, labels, cls_segms))
        return out

    def loss(self, *args, **kwargs):
        raise NotImplementedError("SipMaskHead.loss (training step, SURVEY row a13) lands with the backward kernels")
