"""Second front-end: the maskrcnn-benchmark variant of SipMask.  `B/` = SipMask-benchmark/.

  B/fcos_core/modeling/rpn/sipmask/sipmask.py:48-190    SipMaskHead  (tower convs WITH bias + GN, relu(scale(bbox_pred)),
                                                          DeformConv with bias in FeatureAlign)
  B/fcos_core/modeling/rpn/sipmask/inference.py:28-236  SipMaskPostProcessor ((location, class) pairs, sqrt score,
                                                          boxlist_ml_nms, top DETECTIONS_PER_IMG, mask assembly)
  B/fcos_core/modeling/backbone/{resnet,fpn}.py         R-50-FPN-RETINANET (same arithmetic as the mmdet config)

The kernels and the launch-plan engine are the mmdet variant's; this module converts a B/ checkpoint
(`convert_state_dict`) and exposes the B/ inference contract without yacs / BoxList: results are dicts with the
BoxList fields (`bbox`, `labels`, `scores`, `mask`).
"""
import re

import torch

from .engine import SipMaskEngine
from .plan_cache import PlanCache

DEFAULTS = dict(num_classes=81, fpn_strides=(8, 16, 32, 64, 128), inference_th=0.05, pre_nms_top_n=1000, nms_th=0.6,
                detections_per_img=100, norm_reg_targets=True, centerness_on_reg=True, num_convs=4)


def convert_state_dict(sd):
    """B/ parameter names -> the mmdet names the engine is built from.  Accepts keys with or without the
    'module.' prefix; unknown keys (anchor generators, losses) are dropped."""
    out = {}
    for k, v in sd.items():
        k = k[7:] if k.startswith("module.") else k
        m = re.match(r"backbone\.body\.stem\.(conv1|bn1)\.(.+)$", k)
        if m:
            out["backbone.%s.%s" % m.groups()] = v
            continue
        m = re.match(r"backbone\.body\.(layer\d\.\d+\.(?:conv\d|bn\d|downsample\.\d))\.(.+)$", k)
        if m:
            out["backbone.%s.%s" % m.groups()] = v
            continue
        m = re.match(r"backbone\.fpn\.fpn_inner(\d)\.(weight|bias)$", k)
        if m:
            out["neck.lateral_convs.%d.conv.%s" % (int(m.group(1)) - 2, m.group(2))] = v
            continue
        m = re.match(r"backbone\.fpn\.fpn_layer(\d)\.(weight|bias)$", k)
        if m:
            out["neck.fpn_convs.%d.conv.%s" % (int(m.group(1)) - 2, m.group(2))] = v
            continue
        m = re.match(r"backbone\.fpn\.top_blocks\.p(6|7)\.(weight|bias)$", k)
        if m:
            out["neck.fpn_convs.%d.conv.%s" % (int(m.group(1)) - 3, m.group(2))] = v
            continue
        m = re.match(r"rpn\.head\.(cls|bbox)_tower\.(\d+)\.(weight|bias)$", k)
        if m:
            kind = "cls" if m.group(1) == "cls" else "reg"
            i, r = divmod(int(m.group(2)), 3)            # Sequential(conv, GroupNorm, ReLU) x N
            if r == 0:
                out["bbox_head.%s_convs.%d.conv.%s" % (kind, i, m.group(3))] = v
            elif r == 1:
                out["bbox_head.%s_convs.%d.gn.%s" % (kind, i, m.group(3))] = v
            continue
        m = re.match(r"rpn\.head\.(cls_logits|bbox_pred|centerness)\.(weight|bias)$", k)
        if m:
            name = dict(cls_logits="fcos_cls", bbox_pred="fcos_reg", centerness="fcos_centerness")[m.group(1)]
            out["bbox_head.%s.%s" % (name, m.group(2))] = v
            continue
        m = re.match(r"rpn\.head\.(scales\.\d+\.scale|feat_align\..+|sip_cof\..+|sip_mask_lat0?\..+)$", k)
        if m:
            out["bbox_head." + m.group(1)] = v.reshape(()) if m.group(1).endswith(".scale") else v
            continue
    return out


class SipMaskBenchmark:
    """SipMaskModule at test time (B/...sipmask.py:193-285) on the static HIP launch plan.

        model = SipMaskBenchmark(checkpoint['model'], depth=50)
        results = model(images, image_sizes=[(h, w)] * B, img_metas=[(ori_w, ori_h)] * B)

    images: float [B,3,H,W] on the device (already normalised and padded to SIZE_DIVISIBILITY 32).  All images of a
    batch must share image_size and original size (one launch plan; batch 1 is always fine).  Each result is a dict:
    bbox [N,4] (network-input coordinates, as the BoxList), labels [N] (1-based), scores [N], mask uint8
    [N,1,ori_h,ori_w]."""

    def __init__(self, state_dict, depth=50, device="cuda", **cfg):
        self.cfg = dict(DEFAULTS)
        self.cfg.update(cfg)
        if not (self.cfg["norm_reg_targets"] and self.cfg["centerness_on_reg"]):
            raise NotImplementedError("only the released configs: NORM_REG_TARGETS and CENTERNESS_ON_REG on "
                                      "(configs/sipmask/sipmask_R_50_FPN_1x.yaml)")
        sd = state_dict if any(k.startswith("bbox_head.") for k in state_dict) else convert_state_dict(state_dict)
        self.sd = {k: v.detach() for k, v in sd.items()}
        self.depth, self.device = depth, device
        self._engines = PlanCache()

    def prepare(self, batch, img_hw, image_size, ori_wh):
        key = (batch, tuple(img_hw), tuple(image_size), tuple(ori_wh))
        def build():
            c = self.cfg
            h, w = image_size
            sf = min(h / ori_wh[1], w / ori_wh[0])          # inference.py:198 (np.minimum of the two ratios)
            return SipMaskEngine(self.sd, batch, img_hw, self.depth, None, c["num_classes"], self.device,
                                 tuple(c["fpn_strides"]), (int(h), int(w), 3), scale_factor=sf,
                                 benchmark=dict(pre_nms_thresh=c["inference_th"], pre_nms_top_n=c["pre_nms_top_n"],
                                                nms_thresh=c["nms_th"], post_top_n=c["detections_per_img"]))
        # self.sd holds the caller's tensors: an in-place update of any of them invalidates the plans
        return self._engines.get(key, list(self.sd.values()), build)

    def __call__(self, images, image_sizes, img_metas):
        b = images.shape[0]
        if len(set(map(tuple, image_sizes))) != 1 or len(set(map(tuple, img_metas))) != 1:
            raise NotImplementedError("a batch must share image size and original size (one launch plan)")
        eng = self.prepare(b, tuple(images.shape[-2:]), tuple(image_sizes[0]), tuple(img_metas[0]))
        r = eng.run(images)
        ori_w, ori_h = img_metas[0]
        nd = r["ndet"].cpu().tolist()
        out = []
        for i in range(b):
            n = nd[i]
            m = r["masks"][i, :n]
            canvas = torch.zeros(n, 1, ori_h, ori_w, dtype=torch.uint8, device=m.device)     # inference.py:211-214
            hh, ww = min(m.shape[1], ori_h), min(m.shape[2], ori_w)
            canvas[:, 0, :hh, :ww] = m[:, :hh, :ww]
            det = r["det_bboxes"][i, :n]
            out.append(dict(bbox=det[:, :4], scores=det[:, 4], labels=r["det_labels"][i, :n] + 1, mask=canvas))
        return out
