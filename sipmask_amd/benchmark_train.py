"""Training side of the maskrcnn-benchmark variant.  `B/` = SipMask-benchmark/.

  SipMaskBenchmarkHead     B/fcos_core/modeling/rpn/sipmask/sipmask.py:48-190: parameter container with the reference's
                           state_dict keys (cls_tower.0.weight, bbox_tower.4.bias, scales.2.scale, ...) and the
                           TRAINING-mode forward (tower convs with bias + GN + ReLU, relu(scale(bbox_pred)) left in
                           stride units, FeatureAlign with a biased DeformConv) on the row-tensor HIP autograd ops
                           (ops_rows.py): one launch per conv over the whole pyramid.
  SipMaskLossComputation   B/fcos_core/modeling/rpn/sipmask/loss.py:109-487 with the released yaml's settings
                           (NORM_REG_TARGETS, centre sampling 1.5, GIoU, focal gamma 2 / alpha .25): target assignment on
                           the device (sm_fcos_target), HIP focal loss, GIoU + centerness BCE in tensor code, the mask
                           loss on the fused HIP kernels after the reference's 0.9-IoU NMS filter (sm_nms); num_pos and
                           the centerness sum are all-reduced when torch.distributed is initialised (loss.py:88-94,372-391).

Targets are plain dicts instead of BoxLists: {'bbox': [G,4] xyxy, 'labels': [G] 1-based long, 'masks': uint8 [G,H,W]}.
Inference of the variant lives in benchmark_variant.py (static launch plan); `convert_state_dict` there maps these
parameter names onto the engine's.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import targets as T
from .ops import DeformConv, Scale, mask_loss, nms, sigmoid_focal_loss
from .registry import HEADS

INF = 100000000
REGRESS_RANGES = ((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF))


class _FeatureAlign(nn.Module):
    """sipmask.py:13-47 (B/): conv_offset 1x1 (4 -> 72, no bias), DeformConv 3x3 WITH bias, GN(32), ReLU"""

    def __init__(self, channels, deformable_groups=4):
        super().__init__()
        self.conv_offset = nn.Conv2d(4, deformable_groups * 18, 1, bias=False)
        self.conv_adaption = DeformConv(channels, channels, kernel_size=3, padding=1, deformable_groups=deformable_groups,
                                        bias=True)
        self.norm = nn.GroupNorm(32, channels)

    def init_weights(self):
        nn.init.normal_(self.conv_offset.weight, std=0.0)
        nn.init.normal_(self.conv_adaption.weight, std=0.01)
        nn.init.constant_(self.conv_adaption.bias, 0)


class SipMaskBenchmarkHead(nn.Module):

    def __init__(self, num_classes=81, in_channels=256, num_convs=4, fpn_strides=(8, 16, 32, 64, 128), prior_prob=0.01):
        super().__init__()
        self.fpn_strides = tuple(fpn_strides)
        c = in_channels
        tower = lambda n: nn.Sequential(*[m for _ in range(n) for m in (nn.Conv2d(c, c, 3, 1, 1, bias=True),
                                                                        nn.GroupNorm(32, c), nn.ReLU())])
        self.cls_tower = tower(num_convs - 1)            # sipmask.py:63-81: one conv fewer than the box tower
        self.bbox_tower = tower(num_convs)
        self.cls_logits = nn.Conv2d(c, num_classes - 1, 3, 1, 1)
        self.bbox_pred = nn.Conv2d(c, 4, 3, 1, 1)
        self.centerness = nn.Conv2d(c, 1, 3, 1, 1)
        self.nc = 32
        self.feat_align = _FeatureAlign(c)
        self.sip_cof = nn.Conv2d(c, self.nc * 4, 3, padding=1)
        self.sip_mask_lat = nn.Conv2d(512, self.nc, 3, padding=1)
        self.sip_mask_lat0 = nn.Conv2d(768, 512, 1, padding=0)
        for mods in (self.cls_tower, self.bbox_tower, self.bbox_pred, self.cls_logits, self.centerness):
            for m in mods.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.normal_(m.weight, std=0.01)
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.cls_logits.bias, -math.log((1 - prior_prob) / prior_prob))
        self.scales = nn.ModuleList([Scale(1.0) for _ in range(5)])
        self.feat_align.init_weights()

    def _load_from_state_dict(self, state_dict, prefix, *args):
        # B/ Scale holds a FloatTensor([v]) (shape [1]); ops.Scale a 0-d tensor
        for i in range(len(self.scales)):
            k = "%sscales.%d.scale" % (prefix, i)
            if k in state_dict and state_dict[k].dim() == 1:
                state_dict[k] = state_dict[k].reshape(())
        super()._load_from_state_dict(state_dict, prefix, *args)

    def _tower_rows(self, x, lv, tower):
        from . import ops_rows as R
        mods = list(tower)
        for i in range(0, len(mods), 3):
            conv, gn = mods[i], mods[i + 1]
            y, _ = R.conv_rows(x, lv, conv.weight, conv.bias, 1, 1)
            x = R.gn_rows(y, lv, gn.weight, gn.bias, gn.num_groups, gn.eps, True)
        return x

    def forward_rows(self, pyr, lv):
        """sipmask.py:142-190 in training mode on the pyramid row tensor (all levels / images per launch).  Returns
        (logits, bbox_reg, centerness, cof_preds, feat_masks): lists of NCHW views + the x4-upsampled basis;
        bbox_reg = relu(scale_l * conv) in units of the level stride (NORM_REG_TARGETS, self.training)."""
        from . import ops as P
        from . import ops_rows as R
        from . import hip_ops as H
        b = lv.batch
        cls_t = self._tower_rows(pyr, lv, self.cls_tower)
        box_t = self._tower_rows(pyr, lv, self.bbox_tower)
        zw = self.bbox_pred.weight.new_zeros(3, *self.bbox_pred.weight.shape[1:])
        w_rc = torch.cat([self.bbox_pred.weight, self.centerness.weight, zw], 0)
        b_rc = torch.cat([self.bbox_pred.bias, self.centerness.bias, self.bbox_pred.bias.new_zeros(3)], 0)
        rc, _ = R.conv_rows(box_t, lv, w_rc, b_rc, 1, 1, out_f32=True)
        seg = [(lv.row0[l], lv.row0[l] + b * h * w, h, w) for l, (h, w) in enumerate(lv.sizes)]
        box_rows = [torch.relu(self.scales[l](rc[r0:r1, :4])) for l, (r0, r1, _, _) in enumerate(seg)]
        offset = R.offset_linear_rows(torch.cat([t.detach() for t in box_rows]).float().contiguous(),
                                      self.feat_align.conv_offset.weight.flatten(1), lv)
        ad = self.feat_align.conv_adaption
        y = R.deform_conv_rows(cls_t, lv, offset, ad.weight, ad.bias, 1, 1, ad.deformable_groups)
        n = self.feat_align.norm
        y = R.gn_rows(y, lv, n.weight, n.bias, n.num_groups, n.eps, True)
        nc = self.cls_logits.weight.shape[0]
        cc, _ = R.conv_rows(y, lv, torch.cat([self.cls_logits.weight, self.sip_cof.weight], 0),
                            torch.cat([self.cls_logits.bias, self.sip_cof.bias], 0), 1, 1, out_f32=True)
        view = lambda t, r0, r1, h, w: t[r0:r1].view(b, h, w, t.shape[1]).permute(0, 3, 1, 2)
        logits = [view(cc[:, :nc], *sg) for sg in seg]
        cof_preds = [view(cc[:, nc:], *sg) for sg in seg]
        centerness = [view(rc[:, 4:5], *sg) for sg in seg]
        bbox_reg = [box_rows[l].view(b, h, w, 4).permute(0, 3, 1, 2) for l, (_, _, h, w) in enumerate(seg)]
        h0, w0 = lv.sizes[0]
        l0 = H.Levels(b, [(h0, w0)])
        fm = R.mask_feat_rows(box_t, lv)
        lat0, _ = R.conv_rows(fm, l0, self.sip_mask_lat0.weight, self.sip_mask_lat0.bias, 1, 0, relu=True)
        lat, _ = R.conv_rows(lat0, l0, self.sip_mask_lat.weight, self.sip_mask_lat.bias, 1, 1, relu=True)
        return logits, bbox_reg, centerness, cof_preds, P.upsample_bilinear(R.rows_to_nchw(lat, b, h0, w0), 4)

    def forward(self, x):
        """x: list of 5 NCHW float feature maps on the device -> the training-mode outputs of sipmask.py:142-190.
        (Inference runs from the static launch plan: benchmark_variant.SipMaskBenchmark.)"""
        from . import ops_rows as R
        from . import hip_ops as H
        if not self.training:
            raise NotImplementedError("eval-mode forward of the B/ head is the launch plan of benchmark_variant.SipMaskBenchmark")
        if not x[0].is_cuda:
            raise NotImplementedError("sipmask_amd ops are HIP-only")
        lv = H.Levels(x[0].shape[0], [tuple(f.shape[-2:]) for f in x])
        return self.forward_rows(torch.cat([R.RowsFromNCHW.apply(f) for f in x]), lv)


def compute_locations(features, strides):
    """sipmask.py:259-285: per level [h*w, 2] (x, y) = stride * index + stride // 2"""
    return T.level_points([tuple(f.shape[-2:]) for f in features], strides, torch.float32, features[0].device)


def giou_loss_sum(pred, target, weight):
    """IOULoss('giou'), B/fcos_core/layers/iou_loss.py:8-53, on (l, t, r, b) distances"""
    pa = (pred[:, 0] + pred[:, 2]) * (pred[:, 1] + pred[:, 3])
    ta = (target[:, 0] + target[:, 2]) * (target[:, 1] + target[:, 3])
    wi = torch.min(pred[:, 0], target[:, 0]) + torch.min(pred[:, 2], target[:, 2])
    hi = torch.min(pred[:, 3], target[:, 3]) + torch.min(pred[:, 1], target[:, 1])
    gw = torch.max(pred[:, 0], target[:, 0]) + torch.max(pred[:, 2], target[:, 2])
    gh = torch.max(pred[:, 3], target[:, 3]) + torch.max(pred[:, 1], target[:, 1])
    ac = gw * gh + 1e-7
    inter = wi * hi
    union = ta + pa - inter
    ious = (inter + 1.0) / (union + 1.0)
    losses = 1 - (ious - (ac - union) / ac)
    return (losses * weight).sum() if float(weight.sum()) > 0 else losses.sum()


def _reduce_sum(t):
    """loss.py:88-94: sum over the ranks of the job (identity on one GPU)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    t = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def _num_gpus():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


@HEADS.register_module
class FCOSSipMaskHead(SipMaskBenchmarkHead):
    """The second registry name north_star gives ("SipMaskHead / FCOSSipMaskHead").  The reference has no class of that
    name; it denotes the FCOS-style head of the maskrcnn-benchmark variant
    (B/fcos_core/modeling/rpn/sipmask/sipmask.py:48-190), so `dict(type='FCOSSipMaskHead', ...)` in an mmdet-style config
    builds that head: the reference's B/ parameter names (cls_tower.0.weight, bbox_pred.bias, ...), conv bias + GN towers,
    relu(scale(bbox_pred)), biased DeformConv in FeatureAlign.  Constructor keywords: B/'s (num_convs, fpn_strides,
    prior_prob -- cfg.MODEL.SIPMASK.NUM_CONVS / FPN_STRIDES / PRIOR_PROB, sipmask.py:56-61) or the mmdet spelling of
    the same quantities (stacked_convs, strides); feat_channels must equal in_channels (B/ has one width)."""

    def __init__(self, num_classes=81, in_channels=256, num_convs=None, fpn_strides=None, prior_prob=0.01,
                 stacked_convs=None, strides=None, feat_channels=None, **kwargs):
        if kwargs:
            raise TypeError("FCOSSipMaskHead: unexpected keyword(s) %s" % sorted(kwargs))
        if num_convs is not None and stacked_convs is not None and num_convs != stacked_convs:
            raise ValueError("num_convs and stacked_convs name the same quantity")
        if fpn_strides is not None and strides is not None and tuple(fpn_strides) != tuple(strides):
            raise ValueError("fpn_strides and strides name the same quantity")
        if feat_channels not in (None, in_channels):
            raise ValueError("the B/ head has one width: feat_channels must equal in_channels")
        nconv = num_convs if num_convs is not None else (stacked_convs if stacked_convs is not None else 4)
        st = fpn_strides if fpn_strides is not None else (strides if strides is not None else (8, 16, 32, 64, 128))
        super().__init__(num_classes, in_channels, nconv, tuple(st), prior_prob)
        self.num_classes, self.strides = num_classes, tuple(st)


class SipMaskLossComputation(object):

    def __init__(self, fpn_strides=(8, 16, 32, 64, 128), center_sampling_radius=1.5, gamma=2.0, alpha=0.25):
        self.fpn_strides, self.radius, self.gamma, self.alpha = tuple(fpn_strides), center_sampling_radius, gamma, alpha

    def __call__(self, locations, box_cls, box_regression, centerness, cof_preds, feat_mask, targets):
        """-> (cls_loss, reg_loss, centerness_loss, loss_mask), loss.py:330-487.  box_regression are the TRAINING-mode
        outputs (stride units)."""
        strides = self.fpn_strides
        N, C = box_cls[0].shape[0], box_cls[0].shape[1]
        dev = box_cls[0].device
        gt_bboxes = [t["bbox"].to(dev).float() for t in targets]
        gt_labels = [t["labels"].to(dev).long() for t in targets]
        lab_lvl, tgt_lvl, lab_img, _, gt_inds = T.fcos_target(locations, strides, REGRESS_RANGES, gt_bboxes, gt_labels,
                                                              self.radius > 0, self.radius)
        flat = lambda ts, c: torch.cat([t.permute(0, 2, 3, 1).reshape(-1, c) for t in ts])
        f_cls, f_reg, f_ctr = flat(box_cls, C), flat(box_regression, 4), flat(centerness, 1).reshape(-1)
        f_lab = torch.cat(lab_lvl)
        f_tgt = torch.cat([t / float(s) for t, s in zip(tgt_lvl, strides)])                 # NORM_REG_TARGETS (:283-285)
        pos = torch.nonzero(f_lab > 0).squeeze(1)
        ngpu = float(_num_gpus())
        num_pos = max(float(_reduce_sum(pos.new_tensor([pos.numel()])).item()) / ngpu, 1.0)
        cls_loss = sigmoid_focal_loss(f_cls, f_lab, self.gamma, self.alpha).sum() / num_pos
        p_reg, p_ctr = f_reg[pos], f_ctr[pos]
        if pos.numel() > 0:
            p_tgt = f_tgt[pos]
            ct = T.centerness_target(p_tgt)
            ct_sum = float(_reduce_sum(ct.sum().detach()).item()) / ngpu
            reg_loss = giou_loss_sum(p_reg, p_tgt, ct) / ct_sum
            centerness_loss = F.binary_cross_entropy_with_logits(p_ctr, ct, reduction="sum") / num_pos
        else:
            reg_loss = p_reg.sum()
            _reduce_sum(p_ctr.new_tensor([0.0]))
            centerness_loss = p_ctr.sum()
        # ---- mask loss (:393-482)
        img_cls = torch.cat([c.permute(0, 2, 3, 1).reshape(N, -1, C) for c in box_cls], 1)
        img_cof = torch.cat([c.permute(0, 2, 3, 1).reshape(N, -1, 128) for c in cof_preds], 1)
        img_reg = torch.cat([(r.detach() * float(s)).permute(0, 2, 3, 1).reshape(N, -1, 4)
                             for r, s in zip(box_regression, strides)], 1)              # decode_for_single_feature_map
        cat_pts = torch.cat(locations)
        loss_mask = 0
        for i in range(N):
            labels = torch.cat([l.flatten() for l in lab_img[i]])
            pi = (labels > 0).nonzero().view(-1)
            bdt = T.distance2bbox(cat_pts[pi], img_reg[i][pi]) / 2
            area = (bdt[:, 2] - bdt[:, 0]) * (bdt[:, 3] - bdt[:, 1])
            keep = area > 1.0
            bdt, idx, pk = bdt[keep], gt_inds[i][keep], pi[keep]
            if bdt.shape[0] == 0:
                continue
            with torch.no_grad():
                score = img_cls[i, pk, labels[pk] - 1].sigmoid()
                wgt = score * T.aligned_iou(gt_bboxes[i][idx] / 2, bdt)
                wgt = wgt / wgt.sum() * len(wgt)                                       # no epsilon here (:452)
                _, k = nms(torch.cat([bdt, score[:, None]], 1), 0.9)                   # _box_nms (:453)
                hm, wm = feat_mask[i].shape[1:]
                gt_new = T.prepare_gt_masks(targets[i]["masks"], hm, wm, dev)
            bdt, wgt, idx, pk = bdt[k], wgt[k], idx[k], pk[k]
            bce = mask_loss(feat_mask[i], img_cof[i][pk], bdt, gt_new, idx)
            pre = bce / (bdt[:, 2] - bdt[:, 0]) / (bdt[:, 3] - bdt[:, 1]) / bdt.shape[0]
            loss_mask = loss_mask + torch.sum(pre * wgt)
        loss_mask = loss_mask / N
        if float(loss_mask) > 1.0:                                                     # :483-484
            loss_mask = loss_mask * 0.5
        return cls_loss, reg_loss, centerness_loss, loss_mask


def make_sipmask_loss_evaluator(cfg=None):
    """loss.py:490-492; cfg: dict with the MODEL.SIPMASK keys that matter here (FPN_STRIDES, CENTER_SAMPLING_RADIUS,
    LOSS_GAMMA, LOSS_ALPHA) or None for the released yaml's values"""
    cfg = cfg or {}
    return SipMaskLossComputation(cfg.get("FPN_STRIDES", (8, 16, 32, 64, 128)), cfg.get("CENTER_SAMPLING_RADIUS", 1.5),
                                  cfg.get("LOSS_GAMMA", 2.0), cfg.get("LOSS_ALPHA", 0.25))


class SipMaskBenchmarkModule(nn.Module):
    """SipMaskModule in training mode (sipmask.py:193-257): head + loss evaluator on caller-provided FPN features."""

    def __init__(self, num_classes=81, in_channels=256, num_convs=4, fpn_strides=(8, 16, 32, 64, 128), loss_cfg=None):
        super().__init__()
        self.head = SipMaskBenchmarkHead(num_classes, in_channels, num_convs, fpn_strides)
        self.loss_evaluator = make_sipmask_loss_evaluator(loss_cfg)
        self.fpn_strides = tuple(fpn_strides)

    def forward(self, features, targets):
        """-> dict(loss_cls, loss_reg, loss_centerness, loss_mask) (sipmask.py:242-252)"""
        box_cls, box_regression, centerness, box_cof, feat_mask = self.head(features)
        locations = compute_locations(features, self.fpn_strides)
        c, r, t, m = self.loss_evaluator(locations, box_cls, box_regression, centerness, box_cof, feat_mask, targets)
        return dict(loss_cls=c, loss_reg=r, loss_centerness=t, loss_mask=m)
