"""FCOS target assignment of SipMaskHead (training row a13): M/mmdet/models/anchor_heads/sipmask_head.py:731-866.

Pure tensor code on whatever device the ground truth lives on (the reference runs the same ATen ops there); the
heavy parts of the loss are the HIP ops (focal loss, fused mask loss).
"""
import torch

INF = 1e8


def level_points(featmap_sizes, strides, dtype=torch.float32, device=None):
    """get_points (:664-695): per level [h*w, 2] (x, y) = cell origin + stride // 2, row-major."""
    out = []
    for (h, w), s in zip(featmap_sizes, strides):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=dtype, device=device) * s,
                                torch.arange(w, dtype=dtype, device=device) * s, indexing="ij")
        out.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), -1) + s // 2)
    return out


def assign_image(points, point_stride, lo, hi, gt_bboxes, gt_labels, center_sampling, radius):
    """fcos_target_single (:773-857).  points [P,2]; point_stride, lo, hi [P] (stride and regress range of each
    point's level).  Returns labels [P] long (0 = background), bbox_targets [P,4] (l,t,r,b), gt_ind [#pos]."""
    P, G = points.shape[0], gt_labels.shape[0]
    if G == 0:
        return gt_labels.new_zeros(P), gt_bboxes.new_zeros((P, 4)), gt_labels.new_zeros(0)
    x, y = points[:, 0:1], points[:, 1:2]                          # [P,1] against [1,G]
    bx1, by1, bx2, by2 = (gt_bboxes[:, k].unsqueeze(0) for k in range(4))
    ltrb = torch.stack((x - bx1, y - by1, bx2 - x, by2 - y), -1)   # [P,G,4]
    if center_sampling:
        # the point must fall in the gt's centre box of half-size radius*stride, clipped to the gt (:801-835)
        cx, cy = (bx1 + bx2) / 2, (by1 + by2) / 2
        r = (point_stride * radius).unsqueeze(1)
        cb = torch.stack((x - torch.max(cx - r, bx1), y - torch.max(cy - r, by1),
                          torch.min(cx + r, bx2) - x, torch.min(cy + r, by2) - y), -1)
        inside = cb.min(-1)[0] > 0
    else:
        inside = ltrb.min(-1)[0] > 0
    far = ltrb.max(-1)[0]
    ok = inside & (far >= lo.unsqueeze(1)) & (far <= hi.unsqueeze(1))   # regress range of the level (:841-844)
    area = ((gt_bboxes[:, 2] - gt_bboxes[:, 0] + 1) * (gt_bboxes[:, 3] - gt_bboxes[:, 1] + 1)).unsqueeze(0)
    area = torch.where(ok, area.expand(P, G), area.new_full((), INF))
    best, idx = area.min(dim=1)                                         # smallest covering gt wins (:848-853)
    labels = torch.where(best == INF, gt_labels.new_zeros(()), gt_labels[idx])
    return labels, ltrb[torch.arange(P, device=points.device), idx], idx[labels > 0]


def _assign_batch_device(points, pstride, lo, hi, gt_bboxes_list, gt_labels_list, center_sampling, radius):
    """the whole batch's assignment in ONE launch of sm_fcos_target (no [S,G,4] broadcast tensors, no per-image
    loop of ATen ops); same return values as [assign_image(...) for every image]"""
    from . import hip_ops as H
    B = len(gt_bboxes_list)
    gmax = max(1, max(int(b.shape[0]) for b in gt_bboxes_list))
    dev = points.device
    boxes = torch.zeros(B, gmax, 4, dtype=torch.float32, device=dev)
    labs = torch.zeros(B, gmax, dtype=torch.int64, device=dev)
    for i, (b, l) in enumerate(zip(gt_bboxes_list, gt_labels_list)):
        if b.shape[0]:
            boxes[i, :b.shape[0]] = b.float()
            labs[i, :b.shape[0]] = l
    ngt = torch.tensor([int(b.shape[0]) for b in gt_bboxes_list], dtype=torch.int32, device=dev)
    labels, targets, gidx = H.fcos_target(points.contiguous(), pstride.contiguous(), lo.contiguous(), hi.contiguous(), boxes,
                                          labs, ngt, center_sampling, radius)
    return [(labels[i], targets[i], gidx[i][labels[i] > 0].long()) for i in range(B)]


def fcos_target(points, strides, regress_ranges, gt_bboxes_list, gt_labels_list, center_sampling=True, radius=1.5):
    """fcos_target (:731-771).  Returns (labels per level [cat over images], bbox_targets per level,
    per-image labels split by level, per-image targets split by level, per-image gt_ind)."""
    nums = [p.shape[0] for p in points]
    cat = torch.cat(points)
    mk = lambda vals: torch.cat([p.new_full((p.shape[0],), float(v)) for p, v in zip(points, vals)])
    pstride, lo, hi = mk(strides), mk([r[0] for r in regress_ranges]), mk([r[1] for r in regress_ranges])
    if cat.is_cuda:
        per = _assign_batch_device(cat, pstride, lo, hi, gt_bboxes_list, gt_labels_list, center_sampling, radius)
    else:
        per = [assign_image(cat, pstride, lo, hi, b, l, center_sampling, radius)
               for b, l in zip(gt_bboxes_list, gt_labels_list)]
    lab_img = [p[0].split(nums, 0) for p in per]
    tgt_img = [p[1].split(nums, 0) for p in per]
    lab_lvl = [torch.cat([li[l] for li in lab_img]) for l in range(len(nums))]
    tgt_lvl = [torch.cat([ti[l] for ti in tgt_img]) for l in range(len(nums))]
    return lab_lvl, tgt_lvl, lab_img, tgt_img, [p[2] for p in per]


def centerness_target(pos_bbox_targets):
    """:859-866"""
    lr = pos_bbox_targets[:, [0, 2]]
    tb = pos_bbox_targets[:, [1, 3]]
    return torch.sqrt((lr.min(-1)[0] / lr.max(-1)[0]) * (tb.min(-1)[0] / tb.max(-1)[0]))


def distance2bbox(points, distance, max_shape=None):
    """M/mmdet/core/bbox/transforms.py:202-224"""
    x1, y1 = points[:, 0] - distance[:, 0], points[:, 1] - distance[:, 1]
    x2, y2 = points[:, 0] + distance[:, 2], points[:, 1] + distance[:, 3]
    if max_shape is not None:
        x1, x2 = x1.clamp(0, max_shape[1] - 1), x2.clamp(0, max_shape[1] - 1)
        y1, y2 = y1.clamp(0, max_shape[0] - 1), y2.clamp(0, max_shape[0] - 1)
    return torch.stack([x1, y1, x2, y2], -1)


def aligned_iou(a, b):
    """bbox_overlaps(is_aligned=True), M/mmdet/core/bbox/geometry.py:57-71 (+1 pixel convention)"""
    lt = torch.max(a[:, :2], b[:, :2])
    rb = torch.min(a[:, 2:], b[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    ov = wh[:, 0] * wh[:, 1]
    aa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    return ov / (aa + ab - ov)


def prepare_gt_masks(gt_masks, hm, wm, device):
    """sipmask_head.py:326-329,429-436: instance masks [G,H,W] (array of 0/1) -> bilinear x0.5 -> pasted top-left
    on the [hm,wm] basis grid -> > 0.5, as uint8 on `device` (the gt operand of the fused mask loss)."""
    import numpy as np
    import torch.nn.functional as F
    # ship the masks as uint8 (4x fewer PCIe bytes, no float conversion on the host), widen on the device
    g = torch.from_numpy(np.ascontiguousarray(gt_masks, dtype=np.uint8)).to(device, non_blocking=True).float()
    g = F.interpolate(g.unsqueeze(0), scale_factor=0.5, mode='bilinear', align_corners=False).squeeze(0)
    out = g.new_zeros(g.shape[0], hm, wm)
    h, w = min(hm, g.shape[1]), min(wm, g.shape[2])
    out[:, :h, :w] = g[:, :h, :w]
    return out.gt(0.5).to(torch.uint8)
