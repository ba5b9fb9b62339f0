"""CPU oracle of the maskrcnn-benchmark variant.  TEST INFRASTRUCTURE ONLY.  `B/` = SipMask-benchmark/.

Restates B/fcos_core/modeling/rpn/sipmask/sipmask.py:142-190 (head forward, test mode), inference.py:66-236
(SipMaskPostProcessor), B/fcos_core/csrc/cuda/ml_nms.cu (same-label greedy NMS, IoU with +1) on B/-named parameters.
Pinning: no reference test, but head_forward and postprocess_single reproduce B/'s own SipMaskHead.forward (eval) and
SipMaskPostProcessor.forward run in the build container, and loss() its SipMaskLossComputation (tests/golden/
ref_vectors.npz sections H_ / I_ / L_; stand-ins for _C.ml_nms / _C.nms, DeformConv, CropSplit: tests/golden/ref_loader.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .model import get_points, FPN_STRIDES


def init_head_state_dict(seed=0, num_classes=81, num_convs=4, prefix="rpn.head."):
    """B/ head parameters (sipmask.py:48-140), seeded; gains chosen like oracle.model.init_state_dict(calibrate=True)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    nrm = lambda *s, std=0.01: torch.empty(*s).normal_(0, std, generator=g)
    for kind, n in (("cls", num_convs - 1), ("bbox", num_convs)):
        for i in range(n):
            sd["%s%s_tower.%d.weight" % (prefix, kind, 3 * i)] = nrm(256, 256, 3, 3, std=0.03)
            sd["%s%s_tower.%d.bias" % (prefix, kind, 3 * i)] = nrm(256, std=0.05)
            sd["%s%s_tower.%d.weight" % (prefix, kind, 3 * i + 1)] = torch.ones(256)
            sd["%s%s_tower.%d.bias" % (prefix, kind, 3 * i + 1)] = torch.zeros(256)
    c = num_classes - 1
    sd[prefix + "cls_logits.weight"] = nrm(c, 256, 3, 3, std=0.08)
    sd[prefix + "cls_logits.bias"] = torch.full((c,), -4.0)
    sd[prefix + "bbox_pred.weight"] = nrm(4, 256, 3, 3, std=0.03)
    sd[prefix + "bbox_pred.bias"] = torch.full((4,), 2.0)
    sd[prefix + "centerness.weight"] = nrm(1, 256, 3, 3)
    sd[prefix + "centerness.bias"] = torch.zeros(1)
    for i in range(5):
        sd[prefix + "scales.%d.scale" % i] = torch.tensor([1.0 + 0.25 * i])      # Scale: FloatTensor([init_value])
    sd[prefix + "feat_align.conv_offset.weight"] = nrm(72, 4, 1, 1, std=0.2)
    sd[prefix + "feat_align.conv_adaption.weight"] = nrm(256, 256, 3, 3, std=0.03)
    sd[prefix + "feat_align.conv_adaption.bias"] = nrm(256, std=0.05)
    sd[prefix + "feat_align.norm.weight"] = torch.ones(256)
    sd[prefix + "feat_align.norm.bias"] = torch.zeros(256)
    sd[prefix + "sip_cof.weight"] = nrm(128, 256, 3, 3, std=0.05)
    sd[prefix + "sip_cof.bias"] = torch.zeros(128)
    sd[prefix + "sip_mask_lat.weight"] = nrm(32, 512, 3, 3, std=0.04)
    sd[prefix + "sip_mask_lat.bias"] = torch.zeros(32)
    sd[prefix + "sip_mask_lat0.weight"] = nrm(512, 768, 1, 1, std=0.04)
    sd[prefix + "sip_mask_lat0.bias"] = torch.zeros(512)
    return sd


def head_forward(sd, feats, strides=FPN_STRIDES, prefix="rpn.head."):
    """sipmask.py:142-190 with NORM_REG_TARGETS and CENTERNESS_ON_REG, eval mode (bbox_pred * stride)."""
    h = prefix
    ntow = lambda kind: sum(1 for k in sd if k.startswith("%s%s_tower." % (h, kind)) and k.endswith(".weight")) // 2

    def tower(x, kind):
        for i in range(ntow(kind)):
            x = F.conv2d(x, sd["%s%s_tower.%d.weight" % (h, kind, 3 * i)], sd["%s%s_tower.%d.bias" % (h, kind, 3 * i)], 1, 1)
            x = F.relu(F.group_norm(x, 32, sd["%s%s_tower.%d.weight" % (h, kind, 3 * i + 1)],
                                    sd["%s%s_tower.%d.bias" % (h, kind, 3 * i + 1)], 1e-5))
        return x
    logits, bbox_reg, ctrs, cofs, fm = [], [], [], [], []
    for l, (x, s) in enumerate(zip(feats, strides)):
        ct, bt = tower(x, "cls"), tower(x, "bbox")
        ctrs.append(F.conv2d(bt, sd[h + "centerness.weight"], sd[h + "centerness.bias"], 1, 1))
        bp = F.relu(sd[h + "scales.%d.scale" % l] * F.conv2d(bt, sd[h + "bbox_pred.weight"], sd[h + "bbox_pred.bias"], 1, 1))
        bbox_reg.append(bp * s)
        off = F.conv2d(bp.detach(), sd[h + "feat_align.conv_offset.weight"])     # FeatureAlign.forward: shape.detach() (sipmask.py:44)
        y = ops.deform_conv(ct, off, sd[h + "feat_align.conv_adaption.weight"], 1, 1, 1, 4) + \
            sd[h + "feat_align.conv_adaption.bias"].view(1, -1, 1, 1)
        y = F.relu(F.group_norm(y, 32, sd[h + "feat_align.norm.weight"], sd[h + "feat_align.norm.bias"], 1e-5))
        logits.append(F.conv2d(y, sd[h + "cls_logits.weight"], sd[h + "cls_logits.bias"], 1, 1))
        cofs.append(F.conv2d(y, sd[h + "sip_cof.weight"], sd[h + "sip_cof.bias"], 1, 1))
        if l < 3:
            fm.append(bt if l == 0 else F.interpolate(bt, scale_factor=2 ** l, mode="bilinear", align_corners=False))
    lat0 = F.relu(F.conv2d(torch.cat(fm, 1), sd[h + "sip_mask_lat0.weight"], sd[h + "sip_mask_lat0.bias"]))
    lat = F.relu(F.conv2d(lat0, sd[h + "sip_mask_lat.weight"], sd[h + "sip_mask_lat.bias"], 1, 1))
    return logits, bbox_reg, ctrs, cofs, F.interpolate(lat, scale_factor=4, mode="bilinear", align_corners=False)


def ml_nms(boxes, scores, labels, thr):
    """ml_nms.cu: sort by score (desc; ties by index), greedy, suppress only equal labels with IoU(+1) > thr;
    returns kept indices ascending."""
    boxes, scores, labels = np.asarray(boxes, np.float32), np.asarray(scores, np.float32), np.asarray(labels)
    f1 = np.float32(1)
    area = (boxes[:, 2] - boxes[:, 0] + f1) * (boxes[:, 3] - boxes[:, 1] + f1)
    order = ops.sort_desc_stable(scores)
    keep = []
    for i in order:
        if keep:
            k = np.asarray(keep)
            k = k[labels[k] == labels[i]]                                  # devIoU returns 0 for different labels
            w = np.maximum(np.minimum(boxes[i, 2], boxes[k, 2]) - np.maximum(boxes[i, 0], boxes[k, 0]) + f1, np.float32(0))
            h = np.maximum(np.minimum(boxes[i, 3], boxes[k, 3]) - np.maximum(boxes[i, 1], boxes[k, 1]) + f1, np.float32(0))
            inter = (w * h).astype(np.float32)
            iou = inter / (area[i] + area[k] - inter)
            if (iou > np.float32(thr)).any():
                continue
        keep.append(int(i))
    return np.array(sorted(keep), dtype=np.int64)


def postprocess_single(logits, bbox_reg, ctrs, cofs, feat_mask, image_size, ori_wh, strides=FPN_STRIDES,
                       pre_nms_thresh=0.05, pre_nms_top_n=1000, nms_thresh=0.6, post_top_n=100):
    """inference.py:66-236 for ONE image: per-level tensors [C,h,w] / [4,h,w] / [1,h,w] / [128,h,w]."""
    sizes = [tuple(t.shape[-2:]) for t in logits]
    pts = get_points(sizes, strides)                       # compute_locations: same grid as mmdet's get_points
    B_, S_, L_, F_ = [], [], [], []
    for cls, reg, ctr, cof, loc in zip(logits, bbox_reg, ctrs, cofs, pts):
        C = cls.shape[0]
        p = ops.sigmoid_ref(cls.permute(1, 2, 0).reshape(-1, C))
        r = reg.permute(1, 2, 0).reshape(-1, 4)
        c = ops.sigmoid_ref(ctr.permute(1, 2, 0).reshape(-1))
        f = cof.permute(1, 2, 0).reshape(-1, 128)
        cand = p > pre_nms_thresh
        n = min(int(cand.sum()), pre_nms_top_n)
        prod = p * c[:, None]
        sc = prod[cand]
        nz = cand.nonzero()
        li, cl = nz[:, 0], nz[:, 1] + 1
        if int(cand.sum()) > n:
            flat = li * C + (cl - 1)                        # tie rule: product desc, then pair index asc
            order = torch.from_numpy(ops.sort_desc_stable(sc.numpy()))[:n]
            sc, li, cl = sc[order], li[order], cl[order]
        h, w = image_size
        box = torch.stack([loc[li, 0] - r[li, 0], loc[li, 1] - r[li, 1], loc[li, 0] + r[li, 2], loc[li, 1] + r[li, 3]], 1)
        box[:, 0].clamp_(0, w - 1); box[:, 1].clamp_(0, h - 1); box[:, 2].clamp_(0, w - 1); box[:, 3].clamp_(0, h - 1)
        B_.append(box); S_.append(torch.sqrt(sc)); L_.append(cl); F_.append(f[li])
    boxes, scores, labels, cf = torch.cat(B_), torch.cat(S_), torch.cat(L_), torch.cat(F_)
    keep = torch.from_numpy(ml_nms(boxes.numpy(), scores.numpy(), labels.numpy(), nms_thresh))
    boxes, scores, labels, cf = boxes[keep], scores[keep], labels[keep], cf[keep]
    if len(keep) > post_top_n > 0:                          # kthvalue rule (:177-186): keeps ties at the threshold
        thr = torch.kthvalue(scores, len(keep) - post_top_n + 1)[0]
        k2 = (scores >= thr).nonzero().squeeze(1)
        boxes, scores, labels, cf = boxes[k2], scores[k2], labels[k2], cf[k2]
    out = dict(bbox=boxes, scores=scores, labels=labels, cofs=cf)
    ori_w, ori_h = ori_wh
    sf = min(image_size[0] / ori_h, image_size[1] / ori_w)
    if boxes.shape[0]:
        m = ops.mask_assemble(feat_mask, cf, boxes, 1.0, None, mask_thr=0.4)      # crop with boxes / 2
        up = F.interpolate(m["pos_masks"].unsqueeze(0), scale_factor=2 / sf, mode="bilinear", align_corners=False).squeeze(0)
        masks = (up > 0.4).to(torch.uint8)
        canvas = torch.zeros(masks.shape[0], ori_h, ori_w, dtype=torch.uint8)
        hh, ww = min(masks.shape[1], ori_h), min(masks.shape[2], ori_w)
        canvas[:, :hh, :ww] = masks[:, :hh, :ww]
        out.update(mask=canvas, up=up)
    return out


# ----------------------------------------------------------------------------------------------------------------
# training loss of the variant (SipMaskLossComputation, B/fcos_core/modeling/rpn/sipmask/loss.py:109-487)
# ----------------------------------------------------------------------------------------------------------------
def focal_loss_sum(logits, labels, gamma=2.0, alpha=0.25):
    """SigmoidFocalLoss.forward -> .sum(), B/fcos_core/layers/sigmoid_focal_loss.py:40-69 (the python formula; labels are
    1-based, 0 = background)"""
    C = logits.shape[1]
    cls = torch.arange(1, C + 1, dtype=labels.dtype).unsqueeze(0)
    t = labels.unsqueeze(1)
    p = torch.sigmoid(logits)
    term1 = (1 - p) ** gamma * torch.log(p)
    term2 = p ** gamma * torch.log(1 - p)
    return (-(t == cls).float() * term1 * alpha - ((t != cls) * (t >= 0)).float() * term2 * (1 - alpha)).sum()


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
    return (losses * weight).sum() if weight.sum() > 0 else losses.sum()


def loss(logits, bbox_reg, ctrs, cofs, feat_masks, gt_bboxes, gt_labels, gt_masks_list, strides=FPN_STRIDES,
         radius=1.5, gamma=2.0, alpha=0.25):
    """SipMaskLossComputation.__call__ (:330-487) with the yaml's settings: NORM_REG_TARGETS (bbox_reg are the
    TRAINING-mode outputs relu(scale * conv), regression targets divided by the level stride), center sampling 1.5,
    GIoU.  Returns dict(loss_cls, loss_reg, loss_centerness, loss_mask) and the per-level labels."""
    from .loss import fcos_target_single, centerness_target, prepare_gt_masks, mask_loss_single, _aligned_iou, REGRESS_RANGES
    sizes = [tuple(c.shape[-2:]) for c in logits]
    points = get_points(sizes, strides)
    nums = [p.shape[0] for p in points]
    N, C = logits[0].shape[0], logits[0].shape[1]
    cat_points = torch.cat(points)
    cat_rr = torch.cat([points[i].new_tensor(REGRESS_RANGES[i])[None].expand_as(points[i]) for i in range(len(points))])
    per_img = [fcos_target_single(gt_bboxes[i], gt_labels[i], cat_points, cat_rr, nums, strides, True, radius) for i in range(N)]
    flat = lambda ts, c: torch.cat([t.permute(0, 2, 3, 1).reshape(-1, c) for t in ts])
    f_cls, f_reg, f_ctr = flat(logits, C), flat(bbox_reg, 4), flat(ctrs, 1).reshape(-1)
    f_lab = torch.cat([torch.cat([per_img[i][0].split(nums)[l] for i in range(N)]) for l in range(len(nums))])
    f_tgt = torch.cat([torch.cat([per_img[i][1].split(nums)[l] for i in range(N)]) / strides[l] for l in range(len(nums))])
    pos = (f_lab > 0).nonzero().squeeze(1)
    num_pos = max(float(pos.numel()), 1.0)
    loss_cls = focal_loss_sum(f_cls, f_lab, gamma, alpha) / num_pos
    if pos.numel() > 0:
        ct = centerness_target(f_tgt[pos])
        loss_reg = giou_loss_sum(f_reg[pos], f_tgt[pos], ct) / ct.sum()
        loss_ctr = F.binary_cross_entropy_with_logits(f_ctr[pos], ct, reduction="sum") / num_pos
    else:
        loss_reg, loss_ctr = f_reg[pos].sum(), f_ctr[pos].sum()
    img_cls = torch.cat([c.permute(0, 2, 3, 1).reshape(N, -1, C) for c in logits], 1)
    img_cof = torch.cat([c.permute(0, 2, 3, 1).reshape(N, -1, 128) for c in cofs], 1)
    loss_mask = 0
    for i in range(N):
        labels, _, idx_gt = per_img[i]
        d = torch.cat([bbox_reg[l][i].permute(1, 2, 0).reshape(-1, 4).detach() * strides[l] for l in range(len(points))])
        boxes = torch.stack([cat_points[:, 0] - d[:, 0], cat_points[:, 1] - d[:, 1],
                             cat_points[:, 0] + d[:, 2], cat_points[:, 1] + d[:, 3]], 1) / 2
        pi = (labels > 0).nonzero().view(-1)
        bdt, cof = boxes[pi], img_cof[i][pi]
        area = (bdt[:, 2] - bdt[:, 0]) * (bdt[:, 3] - bdt[:, 1])
        keep = area > 1.0
        bdt, idx, cof = bdt[keep], idx_gt[keep], cof[keep]
        if bdt.shape[0] == 0:
            continue
        score = img_cls[i, pi, labels[pi] - 1].sigmoid().detach()[keep]
        wgt = score * _aligned_iou(gt_bboxes[i][idx] / 2, bdt)
        wgt = wgt / wgt.sum() * len(wgt)                                             # no epsilon here (:452)
        k = torch.from_numpy(np.asarray(ops.nms(torch.cat([bdt, score[:, None]], 1).numpy(), 0.9, mode="gpu"), np.int64))  # :453
        bdt, wgt, idx, cof = bdt[k], wgt[k], idx[k], cof[k]
        hm, wm = feat_masks[i].shape[1:]
        gtm = prepare_gt_masks(gt_masks_list[i], hm, wm)
        li, _ = mask_loss_single(feat_masks[i], cof, bdt, gtm, idx, wgt)
        loss_mask = loss_mask + li
    loss_mask = loss_mask / N
    if float(loss_mask) > 1.0:                                                        # :483-484
        loss_mask = loss_mask * 0.5
    return dict(loss_cls=loss_cls, loss_reg=loss_reg, loss_centerness=loss_ctr, loss_mask=loss_mask), f_lab
