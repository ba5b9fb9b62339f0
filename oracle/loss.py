"""CPU oracle of SipMaskHead.loss (training row a13-a15).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates M/mmdet/models/anchor_heads/sipmask_head.py:289-498 (loss), :731-866 (fcos_target, centerness_target)
with plain differentiable torch-CPU ops, so tests can take both the loss values and, through autograd, the
gradients w.r.t. the head outputs.  Pinning: the reference holds no test or golden value for its loss, but its own `SipMaskHead.loss` / `fcos_target`
run in the build container (with and without center sampling) give the committed values of
tests/golden/ref_vectors.npz section C_, which head_loss reproduces (labels, targets and gt indices exactly, the four
loss values to 2e-5; tests/test_reference_vectors.py).  The CropSplit / CropSplitGt index math comes from oracle.ops (_crop_cells, CUDA-kernel semantics).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .model import get_points, FPN_STRIDES

INF = 1e8
REGRESS_RANGES = ((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF))


def fcos_target_single(gt_bboxes, gt_labels, points, regress_ranges, num_points_per_lvl, strides=FPN_STRIDES,
                       center_sampling=True, radius=1.5):
    """sipmask_head.py:773-857 for one image.  points [P,2], regress_ranges [P,2] ->
    (labels [P] long, bbox_targets [P,4] (l,t,r,b), gt_ind [#pos] long)."""
    P, G = points.shape[0], gt_labels.shape[0]
    if G == 0:      # the reference returns 2 values here and breaks its own caller (:779-781); defined behaviour:
        return gt_labels.new_zeros(P), gt_bboxes.new_zeros((P, 4)), gt_labels.new_zeros(0)
    areas = (gt_bboxes[:, 2] - gt_bboxes[:, 0] + 1) * (gt_bboxes[:, 3] - gt_bboxes[:, 1] + 1)
    areas = areas[None].repeat(P, 1)
    rr = regress_ranges[:, None, :].expand(P, G, 2)
    gb = gt_bboxes[None].expand(P, G, 4)
    xs = points[:, 0][:, None].expand(P, G)
    ys = points[:, 1][:, None].expand(P, G)
    targets = torch.stack((xs - gb[..., 0], ys - gb[..., 1], gb[..., 2] - xs, gb[..., 3] - ys), -1)
    if center_sampling:                                                       # :801-835
        cx = (gb[..., 0] + gb[..., 2]) / 2
        cy = (gb[..., 1] + gb[..., 3]) / 2
        st = cx.new_zeros(cx.shape)
        b = 0
        for lvl, n in enumerate(num_points_per_lvl):
            st[b:b + n] = strides[lvl] * radius
            b += n
        x0 = torch.max(cx - st, gb[..., 0])
        y0 = torch.max(cy - st, gb[..., 1])
        x1 = torch.min(cx + st, gb[..., 2])
        y1 = torch.min(cy + st, gb[..., 3])
        inside = torch.stack((xs - x0, ys - y0, x1 - xs, y1 - ys), -1).min(-1)[0] > 0
    else:
        inside = targets.min(-1)[0] > 0
    mx = targets.max(-1)[0]
    in_range = (mx >= rr[..., 0]) & (mx <= rr[..., 1])                         # :841-844
    areas[~inside] = INF
    areas[~in_range] = INF
    min_area, idx = areas.min(dim=1)                                           # smallest gt wins (:848-853)
    labels = gt_labels[idx].clone()
    labels[min_area == INF] = 0
    return labels, targets[torch.arange(P), idx], idx[labels > 0]


def centerness_target(t):
    """:859-866"""
    lr, tb = t[:, [0, 2]], t[:, [1, 3]]
    return torch.sqrt((lr.min(-1)[0] / lr.max(-1)[0]) * (tb.min(-1)[0] / tb.max(-1)[0]))


def _d2b(points, d):
    """distance2bbox without clamping, transforms.py:202-224"""
    return torch.stack((points[:, 0] - d[:, 0], points[:, 1] - d[:, 1], points[:, 0] + d[:, 2], points[:, 1] + d[:, 3]), -1)


def _aligned_iou(a, b):
    """bbox_overlaps(is_aligned=True), geometry.py:57-71 (+1 convention)"""
    lt = torch.max(a[:, :2], b[:, :2])
    rb = torch.min(a[:, 2:], b[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    ov = wh[:, 0] * wh[:, 1]
    return ov / ((a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1) + (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1) - ov)


def prepare_gt_masks(gt_masks, hm, wm):
    """:326-329,429-436: float masks -> bilinear x0.5 -> top-left paste on the [hm,wm] grid -> > 0.5.
    gt_masks: uint8 array [G,H,W] -> float tensor [G,hm,wm] of 0/1."""
    g = torch.from_numpy(np.array(gt_masks, dtype=np.float32))
    g = F.interpolate(g.unsqueeze(0), scale_factor=0.5, mode="bilinear", align_corners=False).squeeze(0)
    out = g.new_zeros(g.shape[0], hm, wm)
    h, w = min(hm, g.shape[1]), min(wm, g.shape[2])
    out[:, :h, :w] = g[:, :h, :w]
    return out.gt(0.5).float()


def mask_loss_single(feat_mask, cof_pred, bbox_dt, gt_mask_new, idx_gt, weighting, return_pred=False):
    """:438-461 for one image.  feat_mask [32,Hm,Wm], cof_pred [N,128], bbox_dt [N,4] (basis-grid boxes),
    gt_mask_new [G,Hm,Wm], idx_gt [N], weighting [N] -> scalar."""
    hm, wm = feat_mask.shape[1:]
    n = cof_pred.shape[0]
    img = feat_mask.permute(1, 2, 0)
    probs = torch.stack([torch.sigmoid(img @ cof_pred[:, 32 * q:32 * (q + 1)].t()) for q in range(4)], 0)  # [4,H,W,N]
    inside, cell = ops._crop_cells(hm, wm, bbox_dt.detach().numpy(), 2)
    sel = torch.from_numpy(inside[None] & (cell[None] == np.arange(4).reshape(4, 1, 1, 1))).float()
    pred = (probs * sel).sum(0)                                                   # CropSplit forward
    gt = gt_mask_new[idx_gt].permute(1, 2, 0) * torch.from_numpy(inside).float()  # CropSplitGt
    pre = F.binary_cross_entropy(pred, gt, reduction="none").sum(dim=(0, 1))
    w = bbox_dt[:, 2] - bbox_dt[:, 0]
    h = bbox_dt[:, 3] - bbox_dt[:, 1]
    pre = pre / w / h / n
    if return_pred:
        return torch.sum(pre * weighting.detach()), pre, pred
    return torch.sum(pre * weighting.detach()), pre


def head_loss(cls_scores, bbox_preds, centernesses, cof_preds, feat_masks, gt_bboxes, gt_labels, gt_masks_list,
              strides=FPN_STRIDES, regress_ranges=REGRESS_RANGES, center_sampling=True, radius=1.5,
              gamma=2.0, alpha=0.25, stride_norm=True, rescoring_sd=None):
    """SipMaskHead.loss, sipmask_head.py:289-498 (rescoring_flag=False), default loss configs
    (FocalLoss gamma 2 alpha .25, IoULoss, sigmoid CrossEntropyLoss; all loss_weight 1)."""
    sizes = [tuple(c.shape[-2:]) for c in cls_scores]
    points = get_points(sizes, strides)
    nums = [p.shape[0] for p in points]
    num_imgs = cls_scores[0].shape[0]
    C = cls_scores[0].shape[1]
    cat_points = torch.cat(points)
    cat_rr = torch.cat([points[i].new_tensor(regress_ranges[i])[None].expand_as(points[i]) for i in range(len(points))])
    per_img = [fcos_target_single(gt_bboxes[i], gt_labels[i], cat_points, cat_rr, nums, strides, center_sampling, radius)
               for i in range(num_imgs)]
    # level-major flattening (:333-352): level l holds image 0 rows, then image 1 rows, ...
    flat = lambda ts, c: torch.cat([t.permute(0, 2, 3, 1).reshape(-1, c) for t in ts])
    f_cls, f_box, f_ctr = flat(cls_scores, C), flat(bbox_preds, 4), flat(centernesses, 1).reshape(-1)
    lab_l = [torch.cat([per_img[i][0].split(nums)[l] for i in range(num_imgs)]) for l in range(len(nums))]
    tgt_l = [torch.cat([per_img[i][1].split(nums)[l] for i in range(num_imgs)]) for l in range(len(nums))]
    f_lab, f_tgt = torch.cat(lab_l), torch.cat(tgt_l)
    f_pts = torch.cat([p.repeat(num_imgs, 1) for p in points])
    f_str = torch.cat([p.new_full((p.shape[0] * num_imgs, 1), float(s)) for p, s in zip(points, strides)])
    pos = f_lab.nonzero().reshape(-1)
    num_pos = len(pos)
    # FocalLoss -> the CUDA op's formula (sigmoid_focal_loss_cuda.cu:24-59), sum / avg_factor (utils.py:26-52)
    loss_cls = ops.sigmoid_focal_loss_forward(f_cls, f_lab, gamma, alpha).sum() / (num_pos + num_imgs)
    p_box, p_ctr = f_box[pos], f_ctr[pos]
    if num_pos > 0:
        p_tgt = f_tgt[pos]
        ctr_t = centerness_target(p_tgt)
        # M/ decodes stride-normalised distances (:372-375); the VIS head decodes them in pixels (V/...:409-411) -- with
        # the +1 IoU convention the two give different losses
        div = f_str[pos] if stride_norm else 1.0
        dec_p = _d2b(f_pts[pos], p_box / div)
        dec_t = _d2b(f_pts[pos], p_tgt / div)
        iou_l = -_aligned_iou(dec_p, dec_t).clamp(min=1e-6).log()
        loss_bbox = (iou_l * ctr_t).sum() / ctr_t.sum()
        loss_ctr = F.binary_cross_entropy_with_logits(p_ctr, ctr_t, reduction="none").mean()
    else:
        loss_bbox, loss_ctr = p_box.sum(), p_ctr.sum()
    # ---- mask loss (:395-461)
    img_cls = torch.cat([c.permute(0, 2, 3, 1).reshape(num_imgs, -1, C) for c in cls_scores], 1)
    img_cof = torch.cat([c.permute(0, 2, 3, 1).reshape(num_imgs, -1, 128) for c in cof_preds], 1)
    loss_mask = 0
    loss_iou, num_iou = 0, 0.1                                                    # :411-413
    aux = []
    for i in range(num_imgs):
        labels, targets, idx_gt = per_img[i]
        boxes = torch.cat([_d2b(points[l], bbox_preds[l][i].permute(1, 2, 0).reshape(-1, 4).detach())
                           for l in range(len(points))]) / 2                      # det_bboxes[i] / 2 (:407)
        pi = (labels > 0).nonzero().view(-1)
        cof, bdt = img_cof[i][pi], boxes[pi, :4]
        area = (bdt[:, 2] - bdt[:, 0]) * (bdt[:, 3] - bdt[:, 1])
        keep = area > 1.0
        bdt, idx, cof = bdt[keep], idx_gt[keep], cof[keep]
        if bdt.shape[0] == 0:
            loss_mask = loss_mask + area.sum() * 0
            aux.append(None)
            continue
        score = img_cls[i, pi, labels[pi] - 1].sigmoid().detach()[keep]
        ious = _aligned_iou(gt_bboxes[i][idx] / 2, bdt)
        with torch.no_grad():
            wgt = score * ious
            wgt = wgt / (wgt.sum() + 0.0001) * len(wgt)
        hm, wm = feat_masks[i].shape[1:]
        gtm = prepare_gt_masks(gt_masks_list[i][:gt_labels[i].shape[0]], hm, wm)
        li, pre, pred = mask_loss_single(feat_masks[i], cof, bdt, gtm, idx, wgt, return_pred=True)
        loss_mask = loss_mask + li
        aux.append(dict(pos_inds=pi[keep], bbox_dt=bdt, idx_gt=idx, weighting=wgt, pre_loss=pre, gt_mask=gtm))
        if rescoring_sd is not None:                                              # SipMask++ rescoring loss (:463-483)
            from .model import mask_rescoring
            pm = pred.detach().permute(2, 0, 1)                                   # [N,Hm,Wm] cropped probabilities
            pos_labels = labels[pi[keep]] - 1
            pred_iou = mask_rescoring(rescoring_sd, pm, pos_labels, torch.ones(pm.shape[0]))
            with torch.no_grad():
                gsel = gtm[idx]                                                   # UNcropped gt masks of the positives
                mp = (pm > 0.4).float()
                inter = (mp * gsel).sum((1, 2))
                gt_area = gsel.sum((1, 2))
                iou_t = inter / (mp.sum((1, 2)) + gt_area - inter + 0.1)
                iou_w = ((iou_t > 0.1) & (iou_t <= 1.0) & (gt_area >= 100)).float()
            loss_iou = loss_iou + (((pred_iou - iou_t) ** 2) * iou_w).sum()      # MSELoss(reduction='sum') with weights
            num_iou = num_iou + iou_w.sum()
    loss_mask = loss_mask / num_imgs
    out = dict(loss_cls=loss_cls, loss_bbox=loss_bbox, loss_centerness=loss_ctr, loss_mask=loss_mask)
    if rescoring_sd is not None:
        out["loss_iou"] = loss_iou * 10 / num_iou
    return out, dict(labels=f_lab, bbox_targets=f_tgt, per_img=per_img, mask_aux=aux, num_pos=num_pos)
