#!/usr/bin/env python
"""Runs the REFERENCE's own Python code (tests/golden/ref_loader.py explains how) on exact, re-creatable inputs
(oracle/fixtures.py) and stores its outputs in tests/golden/ref_vectors.npz.  Run in the build container only:

    python tests/golden/make_reference_vectors.py

tests/test_reference_vectors.py then checks the oracle against these outputs (no /root/reference needed).
Key prefixes: A_ head forward, B_ get_bboxes, C_ loss / targets, D_ small functions, E_ fast_nms.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader as R  # noqa: E402
from oracle import fixtures as FX  # noqa: E402
from oracle import model as OM  # noqa: E402

NUM_CLASSES = 9          # 8 foreground classes: keeps the fixtures small, exercises the same code
OUT = {}


class Cfg(dict):
    """attribute access like mmcv.Config"""
    __getattr__ = dict.__getitem__


TEST_CFG = Cfg(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=Cfg(type="nms", iou_thr=0.5), max_per_img=100)


def sample_idx(n, k=48):
    return np.unique(np.linspace(0, n - 1, k).astype(np.int64))


def summarize(name, t):
    a = t.detach().cpu().numpy().astype(np.float32).reshape(-1)
    OUT[name + ".sum"] = np.float64(a.astype(np.float64).sum())
    OUT[name + ".abssum"] = np.float64(np.abs(a.astype(np.float64)).sum())
    OUT[name + ".samples"] = a[sample_idx(a.size)]
    OUT[name + ".shape"] = np.asarray(t.shape, np.int64)


def build_head(ns, stacked_convs=4, norm=True, ssd_flag=False, center_sampling=True, rescoring=False):
    kw = dict(num_classes=NUM_CLASSES, in_channels=256, stacked_convs=stacked_convs, feat_channels=256,
              strides=[8, 16, 32, 64, 128], center_sampling=center_sampling, center_sample_radius=1.5, ssd_flag=ssd_flag,
              rescoring_flag=rescoring)
    if not norm:
        kw["norm_cfg"] = None
    head = ns.head.SipMaskHead(**kw)
    tmpl = {k[len("bbox_head."):]: v for k, v in
            OM.init_state_dict(50, 0, num_classes=NUM_CLASSES, stacked_convs=stacked_convs, norm=norm,
                               rescoring=rescoring).items()
            if k.startswith("bbox_head.")}
    assert set(tmpl) == set(head.state_dict()), sorted(set(tmpl) ^ set(head.state_dict()))
    head.load_state_dict(FX.head_state_dict(tmpl))
    head.eval()
    return head


def pack_masks(segms, labels, H, W):
    """the stubbed mask_util.encode returns the pasted uint8 mask: re-order the per-class lists into detection order"""
    cnt, out = {}, []
    for lab in labels.tolist():
        k = cnt.get(lab, 0)
        cnt[lab] = k + 1
        m = np.asarray(segms[lab][k])[:, :, 0]
        assert m.shape == (H, W), m.shape
        out.append(np.packbits(m.astype(np.uint8).reshape(-1)))
    return np.stack(out) if out else np.zeros((0, (H * W + 7) // 8), np.uint8)


def _accept_ndarray_scale_factor():
    """sipmask_head.py:630 passes a numpy array as F.interpolate's scale_factor (fine for the PyTorch of its time);
    today's F.interpolate wants python floats -- convert, change nothing else"""
    import torch.nn.functional as F
    orig = F.interpolate

    def interpolate(input, size=None, scale_factor=None, *a, **k):
        if isinstance(scale_factor, np.ndarray):
            scale_factor = [float(v) for v in scale_factor]
        return orig(input, size, scale_factor, *a, **k)

    F.interpolate = interpolate
    R.STAND_INS["F.interpolate(scale_factor=ndarray)"] = "numpy scale factors converted to python floats (API change of PyTorch)"


def trunk_sections(ns):
    """ResNet-50 caffe style (M/mmdet/models/backbones/resnet.py), plain and with DCN in stages 2-4 (the SipMask++
    configs), and the FPN of the SipMask configs (M/mmdet/models/necks/fpn.py), eval mode"""
    img = FX.texact(601, (1, 3, 64, 96), -2 ** 11, 2 ** 11, 2.0 ** -10)
    for tag, dcn in (("J_backbone", None), ("J_backbone_dcn", (False, True, True, True))):
        kw = dict(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                  norm_cfg=dict(type="BN", requires_grad=False), norm_eval=True, style="caffe")
        if dcn:
            kw.update(dcn=dict(type="DCN", deformable_groups=1, fallback_on_stride=False), stage_with_dcn=dcn)
        net = ns.resnet.ResNet(**kw)
        full = OM.init_state_dict(50, 0, stage_with_dcn=dcn) if dcn else OM.init_state_dict(50, 0)
        tmpl = {k[len("backbone."):]: v for k, v in full.items() if k.startswith("backbone.")}
        assert set(tmpl) == set(net.state_dict())
        net.load_state_dict(FX.trunk_state_dict(tmpl))
        net.eval()
        with torch.no_grad():
            feats = net(img)
        for i, f in enumerate(feats):
            summarize("%s.c%d" % (tag, i + 2), f)
        if dcn is None:
            neck = ns.fpn.FPN(in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1, add_extra_convs=True,
                              extra_convs_on_inputs=False, num_outs=5, relu_before_extra_convs=True)
            ntmpl = {k[len("neck."):]: v for k, v in full.items() if k.startswith("neck.")}
            assert set(ntmpl) == set(neck.state_dict())
            neck.load_state_dict(FX.trunk_state_dict(ntmpl, seed=900))
            neck.eval()
            with torch.no_grad():
                pyr = neck(feats)
            for i, f in enumerate(pyr):
                summarize("J_fpn.p%d" % (i + 3), f)


VIS_CLASSES = 5
VIS_CFG = Cfg(nms_pre=200, min_bbox_size=0, score_thr=0.1, nms=Cfg(type="nms", iou_thr=0.5), max_per_img=10)


def vis_sections():
    """SipMask-VIS head (V/mmdet/models/anchor_heads/sipmask_head.py): forward with the track branch (:252-317),
    get_bboxes with fast_nms and the frame-to-frame matching (:565-684, :544-562, :768-781)"""
    from oracle import vis as OV
    ns = R.mmdet_tree_vis()
    H, W = FX.IMG_HW
    head = ns.head.SipMaskHead(num_classes=VIS_CLASSES, in_channels=256, stacked_convs=3, feat_channels=256,
                               strides=[8, 16, 32, 64, 128], center_sampling=True, center_sample_radius=1.5)
    tmpl = {k[len("bbox_head."):]: v for k, v in OV.init_vis_state_dict(0, num_classes=VIS_CLASSES).items()
            if k.startswith("bbox_head.")}
    assert set(tmpl) == set(head.state_dict())
    head.load_state_dict(FX.head_state_dict(tmpl, seed=300))
    head.eval()
    feats = FX.pyramid_feats(61, 1)
    with torch.no_grad():
        cls, box, ctr, cof, fm, tf, _ = head(feats, feats, flag_train=False)
    for l in range(5):
        summarize("F_vis_forward.cls%d" % l, cls[l])
        summarize("F_vis_forward.box%d" % l, box[l])
        summarize("F_vis_forward.cof%d" % l, cof[l])
    summarize("F_vis_forward.feat_mask", fm)
    summarize("F_vis_forward.track", tf)
    # a 4-frame clip; the reference keeps the tracker memory on the head object
    for case, rescale, sf, ori in (("G_vis_clip", False, 1.0, (H, W, 3)), ("G_vis_clip_rescale", True, 1.5, (64, 85, 3))):
        head.prev_bboxes = head.prev_roi_feats = head.prev_det_labels = None
        for f in range(4):
            cls, box, ctr, cof, fm = FX.head_outputs(71 + (f // 2), 1, VIS_CLASSES - 1)    # frames 0,1 and 2,3 share detections
            box = [b + 0.5 * f for b in box]
            tf = FX.texact(81 + f, (1, 512, H // 8, W // 8), -2 ** 9, 2 ** 9, 2.0 ** -10)
            meta = [dict(img_shape=(H, W, 3), ori_shape=ori, scale_factor=sf, is_first=(f == 0))]
            with torch.no_grad():
                res = head.get_bboxes(cls, box, ctr, cof, fm, tf, tf, meta, VIS_CFG, rescale=rescale)
            det, lab, segms, ids = res[0]
            tag = "%s.f%d" % (case, f)
            OUT[tag + ".det"] = det.numpy().astype(np.float32)
            OUT[tag + ".lab"] = lab.numpy().astype(np.int64)
            OUT[tag + ".ids"] = np.asarray(ids, np.int64)
            keys = sorted(int(k) for k in segms)
            OUT[tag + ".mask_ids"] = np.asarray(keys, np.int64)
            OUT[tag + ".masks"] = (np.stack([np.packbits(np.asarray(segms[k])[:, :, 0].reshape(-1)) for k in keys])
                                   if keys else np.zeros((0, 1), np.uint8))
        OUT[case + ".mask_hw"] = np.asarray(ori[:2], np.int64)
        OUT[case + ".memory_boxes"] = head.prev_bboxes.numpy().astype(np.float32)
        OUT[case + ".memory_labels"] = head.prev_det_labels.numpy().astype(np.int64)

    # loss with the matching term (V/...:320-541).  The reference jitters the reference-frame boxes with
    # Tensor.uniform_ from the global RNG (:470-473): record what it drew, the oracle takes the jitter as an input.
    cls, box, ctr, cof, fm = FX.head_outputs(111, 2, VIS_CLASSES - 1)
    box = [torch.cat([b[:, :2], b[:, :2]], 1) for b in box]    # l = r, t = b: predicted centres stay on the track map
    cof = [c * 0.25 for c in cof]
    gtb, gtl, gtm = FX.ground_truth(112, 2, VIS_CLASSES - 1)
    tf = FX.texact(113, (2, 512, H // 8, W // 8), -2 ** 9, 2 ** 9, 2.0 ** -10)
    tfr = FX.texact(114, (2, 512, H // 8, W // 8), -2 ** 9, 2 ** 9, 2.0 ** -10)
    refb = [b + 2.0 for b in gtb]
    pids = [torch.from_numpy(np.random.RandomState(115 + i).randint(0, len(b) + 1, size=len(b)).astype(np.int64))
            for i, b in enumerate(gtb)]
    drawn = []
    orig_uniform = torch.Tensor.uniform_

    def recording_uniform(self, *a, **k):
        r = orig_uniform(self, *a, **k)
        drawn.append(r.clone().numpy())
        return r

    torch.Tensor.uniform_ = recording_uniform
    torch.manual_seed(7)
    try:
        losses = head.loss(cls, box, ctr, cof, fm * 0.25, tf, tfr, gtb, gtl, [dict(img_shape=(H, W, 3))] * 2, None,
                           gt_masks_list=gtm, ref_bboxes_list=refb, gt_pids_list=pids)
    finally:
        torch.Tensor.uniform_ = orig_uniform
    for k, v in losses.items():
        OUT["K_vis_loss.%s" % k] = np.float64(float(v))
    assert len(drawn) == 2
    for i, d in enumerate(drawn):
        OUT["K_vis_loss.jitter%d" % i] = d.astype(np.float32)


def benchmark_sections():
    """maskrcnn-benchmark variant (B/fcos_core/modeling/rpn/sipmask): head forward in eval mode (sipmask.py:145-190),
    SipMaskPostProcessor.forward (inference.py:66-236) incl. compute_locations (sipmask.py:261-285)"""
    import types
    from oracle import fcos_core as OB
    ns = R.fcos_tree()
    H, W = FX.IMG_HW
    cfg = R.fcos_cfg(num_classes=NUM_CLASSES)
    head = ns.sipmask.SipMaskHead(cfg, 256)
    tmpl = {k[len("rpn.head."):]: v for k, v in OB.init_head_state_dict(0, num_classes=NUM_CLASSES).items()}
    assert set(tmpl) == set(head.state_dict())
    sd = FX.head_state_dict(tmpl, seed=500)
    sd["bbox_pred.bias"] = sd["bbox_pred.bias"] + 1.0          # keep relu(bbox_pred) alive
    head.load_state_dict(sd)
    head.eval()
    feats = FX.pyramid_feats(91, 2)
    with torch.no_grad():
        logits, reg, ctr, cof, fm = head(feats)
    for l in range(5):
        summarize("H_b_forward.cls%d" % l, logits[l])
        summarize("H_b_forward.box%d" % l, reg[l])
        summarize("H_b_forward.ctr%d" % l, ctr[l])
        summarize("H_b_forward.cof%d" % l, cof[l])
    summarize("H_b_forward.feat_mask", fm)

    # inference.py:75-91 calls .view() on the result of permute().reshape().sigmoid(); elementwise results were
    # contiguous in the PyTorch of its time and keep the (non-viewable) input strides today -> same values, contiguous
    _sig = torch.Tensor.sigmoid
    torch.Tensor.sigmoid = lambda self: _sig(self).contiguous()
    R.STAND_INS["Tensor.sigmoid (B/ section)"] = "result made contiguous (memory-format drift of PyTorch; values unchanged)"
    post = ns.inference.SipMaskPostProcessor(pre_nms_thresh=0.05, pre_nms_top_n=1000, nms_thresh=0.6, fpn_post_nms_top_n=100,
                                             min_size=0, num_classes=NUM_CLASSES)
    fake = types.SimpleNamespace(fpn_strides=[8, 16, 32, 64, 128])
    fake.compute_locations_per_level = lambda h, w, s, d: ns.sipmask.SipMaskModule.compute_locations_per_level(fake, h, w, s, d)
    cls, box, ctr, cof, fm = FX.head_outputs(95, 2, NUM_CLASSES - 1)
    locations = ns.sipmask.SipMaskModule.compute_locations(fake, cls)
    OUT["I_b_post.locations"] = torch.cat(locations).numpy()
    for tag, ori_wh in (("I_b_post_same", (W, H)), ("I_b_post_rescale", (85, 64))):
        with torch.no_grad():
            res = post(locations, cls, box, ctr, cof, fm, [(H, W), (H, W)], img_metas=[ori_wh, ori_wh])
        for b, bl in enumerate(res):
            OUT["%s.bbox%d" % (tag, b)] = bl.bbox.numpy().astype(np.float32)
            OUT["%s.scores%d" % (tag, b)] = bl.get_field("scores").numpy().astype(np.float32)
            OUT["%s.labels%d" % (tag, b)] = bl.get_field("labels").numpy().astype(np.int64)
            m = bl.get_field("mask").numpy()[:, 0]
            OUT["%s.masks%d" % (tag, b)] = np.packbits(m.reshape(m.shape[0], -1).astype(np.uint8), axis=1)
        OUT[tag + ".mask_hw"] = np.asarray([ori_wh[1], ori_wh[0]], np.int64)
    torch.Tensor.sigmoid = _sig

    # training loss of the variant (loss.py:330-487).  The CPU branch of SigmoidFocalLoss indexes gamma[0] / alpha[0]
    # (sigmoid_focal_loss.py:43-44), so the cfg carries them as one-element lists; targets are BoxLists whose "masks"
    # field only has to offer get_mask_tensor() (the real SegmentationMask needs cv2 / pycocotools).
    cfg = R.fcos_cfg(num_classes=NUM_CLASSES, LOSS_GAMMA=[2.0], LOSS_ALPHA=[0.25])
    R.STAND_INS["fcos_core SegmentationMask"] = "absent deps: a holder with get_mask_tensor() returning the uint8 masks"
    evaluator = ns.loss.SipMaskLossComputation(cfg)
    cls, box, ctr, cof, fm = FX.head_outputs(97, 2, NUM_CLASSES - 1)
    box = [b / s for b, s in zip(box, FX.STRIDES)]             # training mode: stride-normalised distances
    cof = [c * 0.25 for c in cof]
    gtb, gtl, gtm = FX.ground_truth(98, 2, NUM_CLASSES - 1)

    class _Masks:
        def __init__(self, m):
            self.m = torch.from_numpy(m)

        def get_mask_tensor(self):
            return self.m

    targets = []
    for b in range(2):
        bl = ns.BoxList(gtb[b], (W, H), mode="xyxy")
        bl.add_field("labels", gtl[b])
        bl.add_field("masks", _Masks(gtm[b]))
        targets.append(bl)
    lc, lr, lct, lm = evaluator(locations, cls, box, ctr, cof, fm * 0.25, targets)
    for k, v in (("loss_cls", lc), ("loss_reg", lr), ("loss_centerness", lct), ("loss_mask", lm)):
        OUT["L_b_loss.%s" % k] = np.float64(float(v))


def main():
    torch.manual_seed(0)
    _accept_ndarray_scale_factor()
    ns = R.mmdet_tree(R.M, "M")
    H, W = FX.IMG_HW

    # ---- A: SipMaskHead.forward (sipmask_head.py:241-287), GN head and the SSD-style head (stacked_convs=2, no norm)
    for tag, kw in (("A_forward_gn", dict()), ("A_forward_ssd", dict(stacked_convs=2, norm=False, ssd_flag=True))):
        head = build_head(ns, **kw)
        feats = FX.pyramid_feats(11, 2)
        with torch.no_grad():
            cls, box, ctr, cof, fm = head(feats)
        for l in range(5):
            summarize("%s.cls%d" % (tag, l), cls[l])
            summarize("%s.box%d" % (tag, l), box[l])
            summarize("%s.ctr%d" % (tag, l), ctr[l])
            summarize("%s.cof%d" % (tag, l), cof[l])
        summarize(tag + ".feat_mask", fm)

    # ---- B: get_bboxes / get_bboxes_single (sipmask_head.py:501-663) on synthetic head outputs
    head = build_head(ns)
    cases = [("B_default", None, 1.0, False, (H, W, 3)),
             ("B_rescale", True, 1.5, False, (64, 85, 3)),
             ("B_ssd", True, np.array([1.25, 1.5, 1.25, 1.5], np.float32), True, (64, 102, 3))]
    for tag, rescale, sf, ssd, ori in cases:
        head.ssd_flag = ssd
        outs = FX.head_outputs(21, 2, NUM_CLASSES - 1)
        metas = [dict(img_shape=(H, W, 3), ori_shape=ori, scale_factor=sf)] * 2
        with torch.no_grad():
            res = head.get_bboxes(*outs, metas, TEST_CFG, rescale=rescale)
        for b, (det, lab, segms) in enumerate(res):
            OUT["%s.det%d" % (tag, b)] = det.numpy().astype(np.float32)
            OUT["%s.lab%d" % (tag, b)] = lab.numpy().astype(np.int64)
            mh, mw = (ori[0], ori[1]) if rescale else (H, W)
            OUT["%s.masks%d" % (tag, b)] = pack_masks(segms, lab, mh, mw)
            OUT["%s.mask_hw" % tag] = np.asarray([mh, mw], np.int64)
    head.ssd_flag = False
    # SipMask++ mask rescoring (:635-643): the SSD-style head with rescoring_flag, scores per detection
    head = build_head(ns, stacked_convs=2, norm=False, ssd_flag=True, rescoring=True)
    big = [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]         # the six stride-2 convs need a >= 127 px basis map
    outs = FX.head_outputs(25, 1, NUM_CLASSES - 1, sizes=big)
    sf = np.array([1.25, 1.5, 1.25, 1.5], np.float32)
    metas = [dict(img_shape=(256, 256, 3), ori_shape=(170, 204, 3), scale_factor=sf)]
    with torch.no_grad():
        res = head.get_bboxes(*outs, metas, TEST_CFG, rescale=True)
    for b, (det, lab, (segms, mscores)) in enumerate(res):
        cnt, ms = {}, []
        for l in lab.tolist():
            k = cnt.get(l, 0)
            cnt[l] = k + 1
            ms.append(float(mscores[l][k]))
        OUT["B_rescoring.det%d" % b] = det.numpy().astype(np.float32)
        OUT["B_rescoring.lab%d" % b] = lab.numpy().astype(np.int64)
        OUT["B_rescoring.mask_scores%d" % b] = np.asarray(ms, np.float32)

    # ---- C: loss (sipmask_head.py:290-498) with fcos_target / centerness_target (:731-866)
    for tag, cs in (("C_loss_cs", True), ("C_loss_nocs", False)):
        head = build_head(ns, center_sampling=cs)
        head.center_sample_radius = 1.5
        head.radius = 1.5
        cls, box, ctr, cof, fm = FX.head_outputs(31, 2, NUM_CLASSES - 1)
        cof = [c * 0.25 for c in cof]
        gtb, gtl, gtm = FX.ground_truth(32, 2, NUM_CLASSES - 1)
        fms = fm * 0.25
        metas = [dict(img_shape=(H, W, 3))] * 2
        losses = head.loss(cls, box, ctr, cof, fms, gtb, gtl, metas, None, gt_masks_list=gtm)
        for k, v in losses.items():
            OUT["%s.%s" % (tag, k)] = np.float64(float(v))
        pts, _ = head.get_points([c.shape[-2:] for c in cls], torch.float32, "cpu")
        labels, bbox_targets, label_list, bbox_targets_list, gt_inds = head.fcos_target(pts, gtb, gtl)
        OUT[tag + ".labels"] = torch.cat(labels).numpy().astype(np.int64)            # level-major, images inside a level
        OUT[tag + ".bbox_targets"] = torch.cat(bbox_targets).numpy().astype(np.float32)
        for b in range(2):
            OUT["%s.gt_inds%d" % (tag, b)] = gt_inds[b].numpy().astype(np.int64)
        OUT[tag + ".points"] = torch.cat(pts).numpy().astype(np.float32)

    # ---- C2: SipMask++ loss with the rescoring term loss_iou (:463-491), SSD-style head, 128x128 basis map
    head = build_head(ns, stacked_convs=2, norm=False, ssd_flag=True, rescoring=True)
    big = [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]
    cls, box, ctr, cof, fm = FX.head_outputs(35, 2, NUM_CLASSES - 1, sizes=big)
    cof = [c * 0.25 for c in cof]
    gtb, gtl, gtm = FX.ground_truth(36, 2, NUM_CLASSES - 1, img_hw=(256, 256), max_gt=4)
    gtb = [b * 1.0 for b in gtb]
    losses = head.loss(cls, box, ctr, cof, fm * 0.25, gtb, gtl, [dict(img_shape=(256, 256, 3))] * 2, None, gt_masks_list=gtm)
    for k, v in losses.items():
        OUT["C_loss_rescoring.%s" % k] = np.float64(float(v))

    # ---- D: small functions
    head = build_head(ns)
    a = torch.from_numpy(np.concatenate([FX.exact(41, (40, 2), 0, 2 ** 11, 2.0 ** -4),
                                         FX.exact(42, (40, 2), 0, 2 ** 11, 2.0 ** -4)], 1))
    a = torch.cat([torch.min(a[:, :2], a[:, 2:]), torch.max(a[:, :2], a[:, 2:])], 1)
    b = torch.from_numpy(np.concatenate([FX.exact(43, (40, 2), 0, 2 ** 11, 2.0 ** -4),
                                         FX.exact(44, (40, 2), 0, 2 ** 11, 2.0 ** -4)], 1))
    b = torch.cat([torch.min(b[:, :2], b[:, 2:]), torch.max(b[:, :2], b[:, 2:])], 1)
    OUT["D_overlaps.full"] = ns.geometry.bbox_overlaps(a, b).numpy()
    OUT["D_overlaps.aligned"] = ns.geometry.bbox_overlaps(a, b, is_aligned=True).numpy()
    pts = torch.from_numpy(FX.exact(45, (64, 2), 0, 2 ** 11, 2.0 ** -4))
    dist = torch.from_numpy(FX.exact(46, (64, 4), -64, 2 ** 11, 2.0 ** -4))
    OUT["D_distance2bbox.plain"] = ns.transforms.distance2bbox(pts, dist).numpy()
    OUT["D_distance2bbox.clamped"] = ns.transforms.distance2bbox(pts, dist, max_shape=(96, 128, 3)).numpy()
    t = torch.from_numpy(FX.exact(47, (50, 4), 1, 2 ** 10, 2.0 ** -4))
    OUT["D_centerness_target"] = head.centerness_target(t).numpy()
    # the reference's python crop_split (:58-105, the commented-out alternative of the CUDA op) on box-aligned cases
    data = torch.from_numpy(FX.exact(48, (4, 24, 32, 12), 0, 2 ** 10, 2.0 ** -10))
    rois = torch.from_numpy(FX.exact(49, (12, 4), 0, 96, 0.25))
    rois = torch.cat([torch.min(rois[:, :2], rois[:, 2:]), torch.max(rois[:, :2], rois[:, 2:]) + 2.0], 1)
    OUT["D_py_crop_split"] = ns.head.crop_split(data[0], data[1], data[2], data[3], rois).numpy()

    # ---- E: fast_nms (sipmask_head.py:868-960)
    boxes = torch.from_numpy(np.concatenate([FX.exact(51, (300, 2), 0, 2 ** 11, 2.0 ** -4),
                                             FX.exact(52, (300, 2), 0, 2 ** 11, 2.0 ** -4)], 1))
    boxes = torch.cat([torch.min(boxes[:, :2], boxes[:, 2:]), torch.max(boxes[:, :2], boxes[:, 2:]) + 1.0], 1)
    scores = torch.from_numpy(FX.exact_unique(53, (8, 300)))
    cofs = torch.from_numpy(FX.exact(54, (300, 128)))
    d, l, m = head.fast_nms(boxes, scores, cofs, iou_threshold=0.5, top_k=200, score_thr=0.6)
    OUT["E_fast_nms.det"] = d.numpy()
    OUT["E_fast_nms.lab"] = l.numpy().astype(np.int64)
    OUT["E_fast_nms.cof_rowsum"] = m.numpy().astype(np.float64).sum(1)

    trunk_sections(ns)
    vis_sections()
    benchmark_sections()

    OUT["meta"] = np.asarray(json.dumps(dict(
        generator="tests/golden/make_reference_vectors.py", reference="JialeCao001/SipMask @ v1 (/root/reference)",
        torch=torch.__version__, num_classes=NUM_CLASSES, stand_ins=R.STAND_INS)))
    path = os.path.join(HERE, "ref_vectors.npz")
    np.savez_compressed(path, **OUT)
    print("wrote", path, os.path.getsize(path), "bytes,", len(OUT), "arrays")


if __name__ == "__main__":
    main()
