"""CPU oracle of the maskrcnn-benchmark variant.  TEST INFRASTRUCTURE ONLY.  `B/` = SipMask-benchmark/.

Restates B/fcos_core/modeling/rpn/sipmask/sipmask.py:142-190 (head forward, test mode), inference.py:66-236
(SipMaskPostProcessor), B/fcos_core/csrc/cuda/ml_nms.cu (same-label greedy NMS, IoU with +1) on B/-named parameters.
Pinning: no reference test, but head_forward and postprocess_single reproduce B/'s own SipMaskHead.forward (eval) and
SipMaskPostProcessor.forward run in the build container (tests/golden/ref_vectors.npz sections H_ / I_; stand-ins for
_C.ml_nms, DeformConv, CropSplit: tests/golden/ref_loader.py).
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
        off = F.conv2d(bp, sd[h + "feat_align.conv_offset.weight"])
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
        p = cls.permute(1, 2, 0).reshape(-1, C).sigmoid()
        r = reg.permute(1, 2, 0).reshape(-1, 4)
        c = ctr.permute(1, 2, 0).reshape(-1).sigmoid()
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
