"""SipMaskHead -- drop-in for the reference ``@HEADS`` entry
(M/mmdet/models/anchor_heads/sipmask_head.py:107-960): same constructor kwargs and defaults
(:109-133), same parameter names, same call protocol
(forward / get_bboxes / loss / init_weights).  The arithmetic runs in the HIP engine.
"""
import numpy as np
import torch
import torch.nn as nn

from . import hip_ops as H
from .fp16 import force_fp32
from .modules import ConvModule, bias_init_with_prob, normal_init
from .ops import CropSplit, CropSplitGt, DeformConv, Scale
from .plan_cache import PlanCache, module_tensors
from .registry import HEADS, build_loss

INF = 1e8


_USE_FLAT = True      # the loss reads the head's flat row matrices (False: re-flatten the per-level views; -0.4 ms per step)


class LevelList(list):
    """The per-level NCHW outputs of the row-tensor training forward, as the reference's list -- plus the row matrix they
    are views of (`flat` [sum_l B*h_l*w_l, C], level-major like the reference's own flattening, sipmask_head.py:333-352)
    and its geometry (`lv`).  `loss` reads `flat` directly: flattening five per-level views again costs five slice
    backward passes per output (a zero-fill of the whole matrix + a copy + an add each: 1.3 ms of a 31 ms step)."""
    flat = None
    lv = None

    @classmethod
    def of(cls, items, flat, lv):
        out = cls(items)
        out.flat, out.lv = flat, lv
        return out


def _flat_rows(ts, c):
    """[rows, c] level-major flattening of a list of NCHW tensors (the LevelList's own matrix when there is one)"""
    if isinstance(ts, LevelList) and ts.flat is not None and ts.flat.shape[1] == c and _USE_FLAT:
        return ts.flat
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, c) for t in ts])


def _image_rows(ts, c, num_imgs):
    """per image: (matrix, row index tensor) such that matrix[index] is that image's [sum_l h_l*w_l, c] level-major block"""
    flat = _flat_rows(ts, c)
    sizes = [tuple(t.shape[-2:]) for t in ts]
    out, r0 = [], 0
    starts = []
    for h, w in sizes:
        starts.append(r0)
        r0 += num_imgs * h * w
    for i in range(num_imgs):
        out.append(torch.cat([torch.arange(st + i * h * w, st + (i + 1) * h * w, device=flat.device)
                              for st, (h, w) in zip(starts, sizes)]))
    return flat, out


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
    bbox_loss_stride_norm = True      # loss_bbox on stride-normalised boxes (M/ sipmask_head.py:372-375)

    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=4, strides=(4, 8, 16, 32, 64),
                 regress_ranges=((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF)), center_sampling=False,
                 center_sample_radius=1.5, ssd_flag=False, rescoring_flag=False,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox=dict(type='IoULoss', loss_weight=1.0),
                 loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 conv_cfg=None, norm_cfg=dict(type='GN', num_groups=32, requires_grad=True)):
        super().__init__()
        self.num_classes = num_classes
        self.fp16_enabled = False            # fp16.wrap_fp16_model switches it on (sipmask_head.py: force_fp32 on loss / get_bboxes)
        self.cls_out_channels = num_classes - 1
        self.in_channels, self.feat_channels, self.stacked_convs = in_channels, feat_channels, stacked_convs
        self.strides, self.regress_ranges = strides, regress_ranges
        from . import losses  # noqa: F401  (registers FocalLoss / IoULoss / CrossEntropyLoss / MSELoss)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.loss_centerness = build_loss(loss_centerness)
        if rescoring_flag:                                             # sipmask_head.py:155-156
            self.loss_iou = build_loss(dict(type='MSELoss', loss_weight=1.0, reduction='sum'))
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        self.fp16_enabled = False
        self.center_sampling, self.center_sample_radius = center_sampling, center_sample_radius
        self.ssd_flag, self.rescoring_flag = ssd_flag, rescoring_flag
        self.nc = 32
        self._engines = PlanCache().attach_invalidation(self)
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
        if self.rescoring_flag:                                        # SipMask++ mask scoring branch (:200-219)
            chans = [1, 16, 16, 16, 32, 64, 128]
            self.convs_scoring = nn.Sequential(*[ConvModule(chans[i], chans[i + 1], 3, stride=2, padding=0, bias=True)
                                                 for i in range(6)])
            self.mask_scoring = nn.Conv2d(128, self.num_classes - 1, 1)
            normal_init(self.mask_scoring, std=0.001)
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
        self._engines.clear()

    # ------------------------------------------------------------------ forward (HIP engine, head mode)
    def _engine(self, batch, sizes, img_shape=None, cfg=None):
        from .engine import SipMaskEngine
        key = (batch, tuple(sizes), tuple(img_shape or ()), repr(cfg))
        def build():
            sd = {"bbox_head." + k: v for k, v in self.state_dict().items()}
            return SipMaskEngine.for_head(sd, batch, sizes, num_classes=self.num_classes, strides=self.strides,
                                          test_cfg=cfg, img_shape=img_shape, ssd_flag=self.ssd_flag)
        return self._engines.get(key, module_tensors(self), build)   # rebuilt when any weight has changed since

    # ------------------------------------------------------------------ forward, training mode (autograd)
    def _tower_train(self, x, convs):
        from . import ops as P
        for m in convs:
            if m.with_norm:
                x = P.group_norm(P.conv2d(x, m.conv.weight, None, 1, 1), m.norm.weight, m.norm.bias, 32, 1e-5, True)
            else:
                x = torch.relu(P.conv2d(x, m.conv.weight, m.conv.bias, 1, 1))
        return x

    def rows_path_ok(self):
        """the row-tensor training graph covers the plain SipMaskHead forward (subclasses that add branches to
        forward_train keep their own graph on top of the NCHW views it returns)"""
        return type(self).forward_train is SipMaskHead.forward_train or getattr(self, "_rows_forward_ok", False)

    def _tower_rows(self, x, lv, convs):
        from . import ops_rows as R
        for m in convs:
            if m.with_norm:
                y, _ = R.conv_rows(x, lv, m.conv.weight, None, 1, 1)
                x = R.gn_rows(y, lv, m.norm.weight, m.norm.bias, m.norm.num_groups, m.norm.eps, True)
            else:
                x, _ = R.conv_rows(x, lv, m.conv.weight, m.conv.bias, 1, 1, relu=True)
        return x

    def forward_rows(self, pyr, lv):
        """SipMaskHead.forward (sipmask_head.py:241-287) on the pyramid row tensor: every tower / predictor conv is ONE
        launch over all levels and images (the five levels share the weights), GroupNorm statistics per (image, level),
        FeatureAlign's deformable conv over the whole pyramid, cls + coefficient predictors as one 208-channel conv.
        Returns the reference's five lists of NCHW tensors (views of the row tensors)."""
        from . import ops as P
        from . import ops_rows as R
        from . import hip_ops as H
        b, nl = lv.batch, len(lv)
        cls_feat = self._tower_rows(pyr, lv, self.cls_convs)
        reg_feat = self._tower_rows(pyr, lv, self.reg_convs)
        # fcos_reg (4) + fcos_centerness (1) as ONE 8-channel conv (3 zero channels): the GEMM wants cout % 8 == 0
        zw = self.fcos_reg.weight.new_zeros(3, *self.fcos_reg.weight.shape[1:])
        w_rc = torch.cat([self.fcos_reg.weight, self.fcos_centerness.weight, zw], 0)
        b_rc = torch.cat([self.fcos_reg.bias, self.fcos_centerness.bias, self.fcos_reg.bias.new_zeros(3)], 0)
        rc, _ = R.conv_rows(reg_feat, lv, w_rc, b_rc, 1, 1, out_f32=True)                    # [rows, 8] f32
        seg = [(lv.row0[l], lv.row0[l] + b * h * w, h, w) for l, (h, w) in enumerate(lv.sizes)]
        box_rows = [self.scales[l](rc[r0:r1, :4]) for l, (r0, r1, _, _) in enumerate(seg)]
        # FeatureAlign (:49-55): offsets = conv_offset (1x1, 4 -> 72, no bias) of the DETACHED box prediction
        # (on the plan's own offset kernel + its deterministic 4 x 72 adjoint: the torch matmul put a hipBLASLt kernel with
        # K = every position on the step, 9x the in-tree kernel's time -- VERDICT r3)
        offset = R.offset_linear_rows(torch.cat([t.detach() for t in box_rows]).float().contiguous(),
                                      self.feat_align.conv_offset.weight.flatten(1), lv)
        ad = self.feat_align.conv_adaption
        y = R.deform_conv_rows(cls_feat, lv, offset, ad.weight, None, ad.padding[0] if isinstance(ad.padding, tuple) else ad.padding,
                               1, ad.deformable_groups, relu=not self.feat_align.flag_norm)
        if self.feat_align.flag_norm:
            n = self.feat_align.norm
            y = R.gn_rows(y, lv, n.weight, n.bias, n.num_groups, n.eps, True)
        nc = self.fcos_cls.weight.shape[0]
        cc, _ = R.conv_rows(y, lv, torch.cat([self.fcos_cls.weight, self.sip_cof.weight], 0),
                            torch.cat([self.fcos_cls.bias, self.sip_cof.bias], 0), 1, 1, out_f32=True)
        view = lambda t, r0, r1, h, w: t[r0:r1].view(b, h, w, t.shape[1]).permute(0, 3, 1, 2)
        cls_flat, cof_flat, ctr_flat = cc[:, :nc], cc[:, nc:], rc[:, 4:5]
        box_flat = torch.cat([box_rows[l].float() * self.strides[l] for l in range(nl)])
        cls_scores = LevelList.of([view(cls_flat, *sg) for sg in seg], cls_flat, lv)
        cof_preds = LevelList.of([view(cof_flat, *sg) for sg in seg], cof_flat, lv)
        centernesses = LevelList.of([view(ctr_flat, *sg) for sg in seg], ctr_flat, lv)
        bbox_preds = LevelList.of([view(box_flat, *sg) for sg in seg], box_flat, lv)
        # mask branch (:266-287): [l0 | up2(l1) | up4(l2)] -> 1x1 (768 -> 512) -> 3x3 (512 -> nc) -> x4
        h0, w0 = lv.sizes[0]
        l0 = H.Levels(b, [(h0, w0)])
        fm = R.mask_feat_rows(reg_feat, lv)
        lat0, _ = R.conv_rows(fm, l0, self.sip_mask_lat0.weight, self.sip_mask_lat0.bias, 1, 0, relu=True)
        lat, _ = R.conv_rows(lat0, l0, self.sip_mask_lat.weight, self.sip_mask_lat.bias, 1, 1, relu=True)
        feat_masks = P.upsample_bilinear(R.rows_to_nchw(lat, b, h0, w0), 4)
        return cls_scores, bbox_preds, centernesses, cof_preds, feat_masks

    def forward_train(self, feats):
        """SipMaskHead.forward (sipmask_head.py:241-287) as a differentiable graph of HIP autograd ops
        (ops.conv2d, ops.deform_conv, ops.group_norm, ops.upsample_bilinear; bf16 operands / f32 accumulation in
        the GEMMs, f32 elsewhere), so that `loss(...)` -> `.backward()` produces parameter gradients on the HIP
        kernels.  Layer by layer with NCHW<->NHWC conversions in every conv: correct, not yet fast (the fused static
        plan is inference-only)."""
        from . import ops as P
        from .modules import _train_rows_enabled
        if _train_rows_enabled(feats[0]) and len(feats) >= 3:
            from . import ops_rows as R
            from . import hip_ops as H
            lv = H.Levels(feats[0].shape[0], [tuple(f.shape[-2:]) for f in feats])
            return self.forward_rows(torch.cat([R.RowsFromNCHW.apply(f) for f in feats]), lv)
        cls_scores, bbox_preds, centernesses, cof_preds, fm = [], [], [], [], []
        # fcos_reg (4) + fcos_centerness (1) as ONE 8-channel conv (3 zero channels): the GEMM wants cout % 8 == 0
        zw = self.fcos_reg.weight.new_zeros(3, *self.fcos_reg.weight.shape[1:])
        w_rc = torch.cat([self.fcos_reg.weight, self.fcos_centerness.weight, zw], 0)
        b_rc = torch.cat([self.fcos_reg.bias, self.fcos_centerness.bias, self.fcos_reg.bias.new_zeros(3)], 0)
        for li, (x, scale, stride) in enumerate(zip(feats, self.scales, self.strides)):
            cls_feat = self._tower_train(x, self.cls_convs)
            reg_feat = self._tower_train(x, self.reg_convs)
            rc = P.conv2d(reg_feat, w_rc, b_rc, 1, 1)
            bbox_pred = scale(rc[:, :4])
            centernesses.append(rc[:, 4:5])
            # FeatureAlign (:49-55): the 1x1 offset conv reads the DETACHED box prediction; 4 -> 72 channels is far
            # below one MFMA tile, so it is a plain [72,4] matmul per position (library GEMM; as an ATen conv its
            # weight gradient fell into MIOpen's naive wrw kernel: 28 % of the first profiled training step)
            offset = torch.einsum('oc,bchw->bohw', self.feat_align.conv_offset.weight.flatten(1), bbox_pred.detach())
            y = P.deform_conv(cls_feat, offset, self.feat_align.conv_adaption.weight, 1, 1, 1, 1, 4)
            if self.feat_align.flag_norm:
                y = P.group_norm(y, self.feat_align.norm.weight, self.feat_align.norm.bias, 32, 1e-5, True)
            else:
                y = torch.relu(y)
            cls_scores.append(P.conv2d(y, self.fcos_cls.weight, self.fcos_cls.bias, 1, 1))
            bbox_preds.append(bbox_pred.float() * stride)
            cof_preds.append(P.conv2d(y, self.sip_cof.weight, self.sip_cof.bias, 1, 1))
            if li < 3:
                fm.append(reg_feat if li == 0 else P.upsample_bilinear(reg_feat, 2 ** li))
        lat0 = torch.relu(P.conv2d(torch.cat(fm, 1), self.sip_mask_lat0.weight, self.sip_mask_lat0.bias, 1, 0))
        lat = torch.relu(P.conv2d(lat0, self.sip_mask_lat.weight, self.sip_mask_lat.bias, 1, 1))
        return cls_scores, bbox_preds, centernesses, cof_preds, P.upsample_bilinear(lat, 4)

    def forward(self, feats):
        """feats: tuple of 5 NCHW float tensors -> (cls_scores, bbox_preds, centernesses, cof_preds, feat_masks).
        In training mode with autograd enabled the differentiable path (forward_train) is taken; otherwise the
        fused static launch plan."""
        if self.training and torch.is_grad_enabled():
            return self.forward_train(feats)
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
                             self.strides, rescale, self.ssd_flag, rescore_sd=self._rescore_sd())
        res = post.run()
        if self.rescoring_flag:      # 5th element: mask_scores [N] = predicted mask IoU x box score (:638-641)
            res = [r + (post.mask_scores[b, :r[0].shape[0]],) for b, r in enumerate(res)]
        return res

    @force_fp32(apply_to=('cls_scores', 'bbox_preds', 'centernesses'))
    def get_bboxes(self, cls_scores, bbox_preds, centernesses, cof_preds, feat_masks, img_metas, cfg, rescale=None):
        """Reference packaging (sipmask_head.py:645-662): list of (det_bboxes, det_labels, cls_segms) with
        cls_segms[label] = list of COCO RLE dicts {'size': [H, W], 'counts': bytes}.  The masks are pasted on the
        ori_shape / img_shape canvas and run-length encoded ON DEVICE (sm_rle_encode), so the per-detection
        `mask.cpu()` + `mask_util.encode` loop of the reference becomes two small D2H copies per batch."""
        from .engine import PostProcessor
        post = PostProcessor(cls_scores, bbox_preds, centernesses, cof_preds, feat_masks, img_metas, cfg,
                             self.strides, rescale, self.ssd_flag, rescore_sd=self._rescore_sd())
        res = post.run()
        # every image is pasted on ITS canvas (sipmask_head.py:645-657: ori_shape when rescaling, else img_shape)
        rles = post.encode_rle([tuple((m['ori_shape'] if rescale else m['img_shape'])[:2]) for m in img_metas])
        out = []
        for b, ((det, labels, _, _), rle) in enumerate(zip(res, rles)):
            cls_segms = [[] for _ in range(self.num_classes - 1)]
            lab_h = labels.cpu().tolist()
            for lab, r in zip(lab_h, rle):
                cls_segms[lab].append(r)
            if self.rescoring_flag:      # (cls_segms, mask_scores) bucketed by class, sipmask_head.py:641-643,659-660
                ms = post.mask_scores[b, :det.shape[0]].cpu().numpy()
                la = np.asarray(lab_h, dtype=np.int64)
                out.append((det, labels, (cls_segms, [ms[la == i] for i in range(self.num_classes - 1)])))
            else:
                out.append((det, labels, cls_segms))
        return out

    def _rescore_sd(self):
        if not self.rescoring_flag:
            return None
        return {k: v.detach() for k, v in self.state_dict().items()
                if k.startswith("convs_scoring.") or k.startswith("mask_scoring.")}

    @force_fp32(apply_to=('cls_scores', 'bbox_preds', 'centernesses'))
    def loss(self, cls_scores, bbox_preds, centernesses, cof_preds, feat_masks, gt_bboxes, gt_labels, img_metas, cfg,
             gt_bboxes_ignore=None, gt_masks_list=None, _per_image=None, _targets=None):
        """sipmask_head.py:289-498: dict(loss_cls, loss_bbox, loss_centerness, loss_mask[, loss_iou]).

        Same arguments as the reference.  Target assignment is tensor code (targets.py); the classification
        loss runs on the HIP sigmoid-focal-loss kernel and the mask loss on the fused HIP kernels
        (ops.mask_loss: no [4,Hm,Wm,N] probability volumes, CropSplit/CropSplitGt/BCE/reduction in one pass),
        both differentiable, so `sum(losses).backward()` reaches the head outputs."""
        from . import targets as T
        from .ops import mask_loss
        if gt_masks_list is None:
            raise ValueError("SipMaskHead.loss needs gt_masks_list (with_mask=True in the train pipeline)")
        assert len(cls_scores) == len(bbox_preds) == len(centernesses)
        dev = cls_scores[0].device
        sizes = [tuple(f.shape[-2:]) for f in cls_scores]
        num_imgs, C = cls_scores[0].size(0), self.cls_out_channels
        tg = _targets
        if tg is None or tg["sizes"] != sizes or tg["mask_hw"] != tuple(feat_masks.shape[-2:]):
            tg = self.prepare_targets(sizes, tuple(feat_masks.shape[-2:]), gt_bboxes, gt_labels, gt_masks_list, dev,
                                      bbox_preds[0].dtype)
        points, f_lab, f_tgt, f_pts, f_str, pos, num_pos = (tg[k] for k in ("points", "f_lab", "f_tgt", "f_pts", "f_str",
                                                                           "pos", "num_pos"))
        f_cls, f_box, f_ctr = _flat_rows(cls_scores, C), _flat_rows(bbox_preds, 4), _flat_rows(centernesses, 1).reshape(-1)
        loss_cls = self.loss_cls(f_cls, f_lab, avg_factor=num_pos + num_imgs)             # :364-366
        p_box, p_ctr = f_box[pos], f_ctr[pos]
        if num_pos > 0:
            p_tgt = f_tgt[pos]
            ctr_t = T.centerness_target(p_tgt)
            # M/ :372-375 decodes stride-normalised distances; the VIS head (V/...:409-411) decodes pixels: with the
            # +1 IoU convention that is a different loss, so the subclass switches it off
            div = f_str[pos] if self.bbox_loss_stride_norm else 1.0
            dec_p = T.distance2bbox(f_pts[pos], p_box / div)
            dec_t = T.distance2bbox(f_pts[pos], p_tgt / div)
            loss_bbox = self.loss_bbox(dec_p, dec_t, weight=ctr_t, avg_factor=ctr_t.sum())   # :379-383
            loss_centerness = self.loss_centerness(p_ctr, ctr_t)
        else:
            loss_bbox, loss_centerness = p_box.sum(), p_ctr.sum()
        # ---- mask loss (:395-461), one fused launch pair per image
        # per-image access to the level-major matrices through row indices (a gather of the positives' rows: one
        # index_put in the backward instead of the cat / permute / slice chain of a per-image regrouping)
        cof_flat, img_rows = _image_rows(cof_preds, 128, num_imgs)
        cls_flat, box_flat = f_cls.detach(), f_box.detach()
        cat_pts = torch.cat(points)
        loss_mask = 0
        loss_iou, num_iou = 0, 0.1                                                        # :404-405
        # The reference drops positives whose predicted box has area <= 1 by boolean indexing (:421-424) -- a
        # data-dependent shape, i.e. a device->host sync per image in the middle of the step.  Without per-image hooks
        # the same sums are formed over ALL positives with the dropped ones masked to zero: no sync after the targets.
        masked = _per_image is None and not self.rescoring_flag
        for i in range(num_imgs):
            labels, pi, gt_new = tg["labels"][i], tg["pi"][i], tg["gt_new"][i]
            if pi.numel() == 0:
                continue
            ri = img_rows[i][pi]                                                          # rows of this image's positives
            bdt = T.distance2bbox(cat_pts[pi], box_flat[ri]) / 2                          # det_bboxes[i] / 2
            wb, hb = bdt[:, 2] - bdt[:, 0], bdt[:, 3] - bdt[:, 1]
            keep = wb * hb > 1.0
            idx = tg["gt_inds"][i]
            if masked:
                with torch.no_grad():
                    cnt = keep.sum()
                    score = cls_flat[ri, labels[pi] - 1].sigmoid()
                    wgt = torch.where(keep, score * T.aligned_iou(gt_bboxes[i][idx] / 2, bdt), score.new_zeros(()))
                    wgt = wgt / (wgt.sum() + 0.0001) * cnt
                    den = torch.where(keep, wb * hb, wb.new_ones(())) * cnt.clamp(min=1)
                bce = mask_loss(feat_masks[i], cof_flat[ri], bdt, gt_new, idx)           # [N] per-detection sums
                loss_mask = loss_mask + torch.sum(torch.where(keep, bce / den, bce.new_zeros(())) * wgt)
                continue
            bdt, idx, pk, rk = bdt[keep], idx[keep], pi[keep], ri[keep]
            if bdt.shape[0] == 0:
                continue
            with torch.no_grad():
                score = cls_flat[rk, labels[pk] - 1].sigmoid()
                weighting = score * T.aligned_iou(gt_bboxes[i][idx] / 2, bdt)
                weighting = weighting / (weighting.sum() + 0.0001) * len(weighting)
            if _per_image is not None:       # hook for heads that add per-image terms (VIS track loss)
                _per_image(i, bdt, idx)
            bce = mask_loss(feat_masks[i], cof_flat[rk], bdt, gt_new, idx)               # [N] per-detection sums
            pre = bce / (bdt[:, 2] - bdt[:, 0]) / (bdt[:, 3] - bdt[:, 1]) / bdt.shape[0]
            loss_mask = loss_mask + torch.sum(pre * weighting)
            if self.rescoring_flag:                                                       # :463-483
                li, wi = self._rescoring_loss(feat_masks[i], cof_flat[rk], bdt, gt_new, idx, labels[pk] - 1)
                loss_iou, num_iou = loss_iou + li, num_iou + wi
        if not torch.is_tensor(loss_mask):
            loss_mask = f_box.sum() * 0                                                   # no positives anywhere (:425-427)
        loss_mask = loss_mask / num_imgs
        out = dict(loss_cls=loss_cls, loss_bbox=loss_bbox, loss_centerness=loss_centerness, loss_mask=loss_mask)
        if self.rescoring_flag:
            out["loss_iou"] = loss_iou * 10 / num_iou                                     # :485-486
        return out

    def prepare_targets(self, sizes, mask_hw, gt_bboxes, gt_labels, gt_masks_list, device, dtype=torch.float32):
        """Everything of `loss` that depends on the ground truth only (sipmask_head.py:311-352, 405-436): point grid, FCOS
        target assignment (one device launch), the positives' indices (the `nonzero()` calls -- host syncs -- of the
        loss) and the ground-truth masks on the basis grid.  `SipMask.forward_train` calls it BEFORE the forward pass,
        so those syncs wait on an idle device instead of draining the forward's launch queue mid-step; `loss` computes
        it itself when it is not handed one."""
        from . import targets as T
        sizes = [tuple(s) for s in sizes]
        points = T.level_points(sizes, self.strides, dtype, device)
        nums = [p.shape[0] for p in points]
        num_imgs = len(gt_bboxes)
        lab_lvl, tgt_lvl, lab_img, _, gt_inds = T.fcos_target(points, self.strides, self.regress_ranges, gt_bboxes, gt_labels,
                                                              self.center_sampling, self.center_sample_radius)
        f_lab, f_tgt = torch.cat(lab_lvl), torch.cat(tgt_lvl)
        pos = f_lab.nonzero().reshape(-1)
        labels = [torch.cat([l.flatten() for l in lab_img[i]]) for i in range(num_imgs)]
        pi = [(lab > 0).nonzero().view(-1) for lab in labels]
        hm, wm = mask_hw
        gt_new = [T.prepare_gt_masks(gt_masks_list[i][:gt_labels[i].shape[0]], hm, wm, device) if pi[i].numel() else None
                  for i in range(num_imgs)]
        return dict(sizes=sizes, mask_hw=(hm, wm), points=points, f_lab=f_lab, f_tgt=f_tgt,
                    f_pts=torch.cat([p.repeat(num_imgs, 1) for p in points]),
                    f_str=torch.cat([p.new_full((n * num_imgs, 1), float(s)) for p, n, s in zip(points, nums, self.strides)]),
                    pos=pos, num_pos=len(pos), labels=labels, pi=pi, gt_inds=gt_inds, gt_new=gt_new)

    def _rescoring_loss(self, feat_mask, cof, bdt, gt_new, idx, pos_labels):
        """SipMask++ rescoring loss of one image (sipmask_head.py:463-483): the DETACHED cropped probability masks of
        the positives go through convs_scoring (six 3x3 stride-2 convs) + mask_scoring + global max; target = IoU of
        the mask thresholded at 0.4 with the (uncropped) ground-truth mask.  The conv chain runs on the row-tensor
        training ops (one 'image' per positive), so gradients reach exactly the scoring branch's parameters."""
        from . import ops as P
        from . import ops_rows as R
        from . import hip_ops as H
        n = bdt.shape[0]
        with torch.no_grad():
            img = feat_mask.detach().float().permute(1, 2, 0)                            # [Hm,Wm,32]
            c = cof.detach().float()
            probs = torch.stack([torch.sigmoid(img @ c[:, 32 * q:32 * (q + 1)].t()) for q in range(4)], 0).contiguous()
            pred = P.crop_split(probs, bdt.detach().float().contiguous(), 2)              # [Hm,Wm,N] cropped probabilities
            gsel = gt_new[idx].float()                                                   # [N,Hm,Wm] UNcropped gt masks
            pm = pred.permute(2, 0, 1)
            mp = (pm > 0.4).float()
            inter = (mp * gsel).sum((1, 2))
            gt_area = gsel.sum((1, 2))
            iou_t = inter / (mp.sum((1, 2)) + gt_area - inter + 0.1)
            iou_w = ((iou_t > 0.1) & (iou_t <= 1.0) & (gt_area >= 100)).float()
            hm, wm = pm.shape[1:]
            x = torch.zeros(n * hm * wm, 8, dtype=torch.bfloat16, device=pm.device)       # 1 channel padded to 8
            x[:, 0] = pm.reshape(-1)
        lv = H.Levels(n, [(hm, wm)])
        for m in self.convs_scoring:
            x, lv = R.conv_rows(x, lv, m.conv.weight, m.conv.bias, 2, 0, relu=True)
        x, lv = R.conv_rows(x, lv, self.mask_scoring.weight, self.mask_scoring.bias, 1, 0, relu=True, out_f32=True)
        h, w = lv.sizes[0]
        pred_iou = x.view(n, h * w, -1).max(1).values[torch.arange(n, device=x.device), pos_labels]
        return self.loss_iou(pred_iou.view(-1, 1), iou_t.view(-1, 1), iou_w.view(-1, 1)), iou_w.sum()
