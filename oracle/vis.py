"""CPU oracle of the SipMask-VIS head.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

`V/` = /root/reference/SipMask-VIS/.  Restates V/mmdet/models/anchor_heads/sipmask_head.py:
  forward with the track branch            :252-317 (track_convs :219-232)
  get_bboxes_single (always fast_nms)      :686-766, fast_nms :951-993
  extract_box_feature_center_single        :768-781
  frame-to-frame matching                  :565-684 (compute_comp_scores :544-562)
Pinning: no reference test, but track_forward, get_masks_single_vis, extract_box_feature_center and Tracker reproduce
the VIS head's own forward / get_bboxes over a 4-frame clip run in the build container (tests/golden/ref_vectors.npz
sections F_ / G_: detections, object ids, memory and every mask pixel), track_loss its loss_match (section K_).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .model import _tower, select_candidates, init_state_dict

MATCH_COEFF = (1.0, 2.0, 10.0)      # :165


def init_vis_state_dict(seed=0, num_classes=41, stacked_convs=3):
    """The M/ head layout plus track_convs.{i}.{conv,gn} and sipmask_track (:219-232)."""
    sd = init_state_dict(50, seed, True, num_classes, stacked_convs, True)
    g = torch.Generator().manual_seed(seed + 77)
    h = "bbox_head."
    for i in range(stacked_convs - 1):
        sd[h + "track_convs.%d.conv.weight" % i] = torch.empty(256, 256, 3, 3).normal_(0, 0.03, generator=g)
        sd[h + "track_convs.%d.gn.weight" % i] = torch.ones(256)
        sd[h + "track_convs.%d.gn.bias" % i] = torch.zeros(256)
    sd[h + "sipmask_track.weight"] = torch.empty(512, 768, 1, 1).uniform_(-0.036, 0.036, generator=g)
    sd[h + "sipmask_track.bias"] = torch.empty(512).uniform_(-0.036, 0.036, generator=g)
    return sd


def track_forward(sd, feats, prefix="bbox_head."):
    """track_feats [B,512,h0,w0] of the test path (flag_train=False), :265-284,310-311."""
    h = prefix
    n = sum(1 for k in sd if k.startswith(h + "track_convs.") and k.endswith(".conv.weight"))
    outs = []
    for li, x in enumerate(feats[:3]):
        t = x
        for i in range(n):
            t = _tower(sd, t, h + "track_convs.%d" % i)
        outs.append(F.interpolate(t, scale_factor=2 ** li, mode="bilinear", align_corners=False))
    return F.conv2d(torch.cat(outs, 1), sd[h + "sipmask_track.weight"], sd[h + "sipmask_track.bias"])


def get_masks_single_vis(cls_scores, bbox_preds, ctrs, cofs, feat_mask, img_shape, cfg, scale_factor=1.0, rescale=False):
    """V/...:686-766: fast_nms over score*centerness (top 200 per class, cfg.score_thr, cfg.max_per_img), crop
    boxes * scale_factor / 2 and upsampling 2 / scale_factor only when rescale, mask threshold 0.5."""
    mb, ms, mc, mf, lv, ps = select_candidates(cls_scores, bbox_preds, ctrs, cofs, img_shape, cfg.get("nms_pre", -1))
    if rescale:
        mb = mb / torch.as_tensor(scale_factor, dtype=torch.float32)
    sc = (ms * mc.view(-1, 1))[:, 1:].t().contiguous()
    det, lab, det_cofs = ops.fast_nms(mb.numpy(), sc.numpy(), mf.numpy(), 0.5, 200, cfg["score_thr"], cfg["max_per_img"])
    out = dict(det_bboxes=det, det_labels=lab)
    if det.shape[0] > 0:
        out.update(ops.mask_assemble(feat_mask, torch.from_numpy(det_cofs), det, scale_factor,
                                     True if rescale else None, mask_thr=0.5))
    return out


def extract_box_feature_center(track_feats, boxes, stride=8):
    """:768-781: the 512-vector at the box centre on the stride-8 track map.  track_feats [512,h,w], boxes [N,4]."""
    boxes = torch.as_tensor(boxes, dtype=torch.float32)
    cx = torch.floor((boxes[:, 2] + boxes[:, 0]) / 2.0 / stride).long()
    cy = torch.floor((boxes[:, 3] + boxes[:, 1]) / 2.0 / stride).long()
    return track_feats.permute(1, 2, 0)[cy, cx, :]


def _overlaps(a, b):
    """bbox_overlaps (V/mmdet/core/bbox/geometry.py, +1 convention), [n,4] x [m,4] -> [n,m]"""
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    ov = wh[..., 0] * wh[..., 1]
    aa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    return ov / (aa[:, None] + ab[None, :] - ov)


def comp_scores(det_feats, prev_feats, det_boxes, det_labels, prev_boxes, prev_labels, coeff=MATCH_COEFF):
    """:623-637 + compute_comp_scores(add_bbox_dummy=True) :544-562 -> [N, T+1] (column 0 = "new object")."""
    prod = det_feats @ prev_feats.t()
    score = torch.cat([prod.new_zeros(prod.shape[0], 1), prod], 1)
    logp = F.log_softmax(score, dim=1)
    delta = (prev_labels.view(1, -1) == det_labels.view(-1, 1)).float()
    ious = _overlaps(det_boxes[:, :4], prev_boxes[:, :4])
    n = prod.shape[0]
    ious = torch.cat([ious.new_zeros(n, 1), ious], 1)
    delta = torch.cat([delta.new_ones(n, 1), delta], 1)
    return logp + coeff[0] * torch.log(det_boxes[:, 4].view(-1, 1)) + coeff[1] * ious + coeff[2] * delta


class Tracker:
    """The per-video state of V/...:169-171 and its update rule :618-667."""

    def __init__(self):
        self.prev_bboxes = self.prev_roi_feats = self.prev_det_labels = None

    def step(self, det_boxes, det_labels, det_feats, is_first):
        """det_boxes [N,5] (as returned by get_bboxes_single), det_labels [N] long, det_feats [N,512].
        Returns det_obj_ids (int array [N], -1 = dropped duplicate)."""
        det_boxes = torch.as_tensor(det_boxes, dtype=torch.float32)
        det_labels = torch.as_tensor(det_labels, dtype=torch.long)
        n = det_boxes.shape[0]
        if is_first or self.prev_bboxes is None:
            self.prev_bboxes, self.prev_roi_feats, self.prev_det_labels = det_boxes.clone(), det_feats.clone(), det_labels.clone()
            return np.arange(n)
        cs = comp_scores(det_feats, self.prev_roi_feats, det_boxes, det_labels, self.prev_bboxes, self.prev_det_labels)
        match_ids = cs.max(dim=1)[1].numpy().astype(np.int32)
        ids = -np.ones(n, np.int32)
        best = -100.0 * np.ones(self.prev_bboxes.shape[0])
        for i, m in enumerate(match_ids):
            if m == 0:                                        # a new object: append to the memory
                ids[i] = self.prev_roi_feats.shape[0]
                self.prev_roi_feats = torch.cat((self.prev_roi_feats, det_feats[i][None]), 0)
                self.prev_bboxes = torch.cat((self.prev_bboxes, det_boxes[i][None]), 0)
                self.prev_det_labels = torch.cat((self.prev_det_labels, det_labels[i][None]), 0)
            else:                                             # several detections may claim one object: best wins
                o = m - 1
                if float(cs[i, m]) > best[o]:
                    ids[i] = o
                    best[o] = float(cs[i, m])
                    self.prev_roi_feats[o] = det_feats[i]
                    self.prev_bboxes[o] = det_boxes[i]
        return ids


def track_loss(track_feats, track_feats_ref, mask_aux, ref_bboxes_list, gt_pids_list, jitter):
    """loss_match of V/...:470-498,536 given the per-image positives of the mask loss (oracle.loss.head_loss aux):
    mean over images of cross_entropy([0, f_key . f_ref^T], gt_pids[idx_gt])."""
    total = 0
    for i, aux in enumerate(mask_aux):
        if aux is None:
            continue
        ref = ref_bboxes_list[i]
        off = jitter[i]
        cxcy = (ref[:, 2:4] + ref[:, :2]) / 2
        wh = (ref[:, 2:4] - ref[:, :2]).abs()
        ncxcy, nwh = cxcy + wh * off[:, :2], wh * (1 + off[:, 2:])
        new_boxes = torch.cat([ncxcy - nwh / 2, ncxcy + nwh / 2], 1)
        fk = extract_box_feature_center(track_feats[i], aux["bbox_dt"] * 2)
        fr = extract_box_feature_center(track_feats_ref[i], new_boxes)
        prod = fk @ fr.t()
        prod_ext = torch.cat([prod.new_zeros(prod.shape[0], 1), prod], 1)
        total = total + F.cross_entropy(prod_ext, gt_pids_list[i][aux["idx_gt"]], reduction="mean")
    return total / len(mask_aux)
