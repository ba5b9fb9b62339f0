"""CPU oracle: the SipMask detector graph as pure functions over a reference-named
``state_dict`` (TEST INFRASTRUCTURE ONLY -- see oracle/ops.py header).

Restates (fp32, torch-CPU ATen convs):
  * ResNet caffe-style bottleneck backbone  M/mmdet/models/backbones/resnet.py:84-239,311-521
  * FPN                                      M/mmdet/models/necks/fpn.py:137-178
  * ConvModule conv->GN->ReLU                M/mmdet/ops/conv_module.py:34-132
  * SipMaskHead.forward / get_bboxes_single  M/mmdet/models/anchor_heads/sipmask_head.py:241-287,543-633
Pinning: the reference has no test on this graph; head_forward and get_masks_single reproduce the
reference's own SipMaskHead.forward / get_bboxes run in the build container (tests/golden/ref_vectors.npz, sections A_
and B_; stand-ins only for DeformConv, CropSplit and NMS, see tests/golden/ref_loader.py); backbone_forward (plain and
DCN) and fpn_forward reproduce the reference's ResNet / FPN modules (section J_), mask_rescoring its rescoring branch.
"""
import math

import torch
import torch.nn.functional as F

from . import ops

ARCH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}
FPN_STRIDES = (8, 16, 32, 64, 128)


# ----------------------------------------------------------------------------
# parameter construction (reference init, then the calibration overrides of
# SURVEY section 8d so that synthetic weights exercise NMS / deform / masks)
# ----------------------------------------------------------------------------


def _kaiming(w, g):   # mmcv kaiming_init: fan_out, relu, normal
    fan_out = w.shape[0] * w.shape[2] * w.shape[3]
    return w.normal_(0, math.sqrt(2.0 / fan_out), generator=g)


def _xavier_uniform(w, g):   # mmcv xavier_init(distribution='uniform'), fpn.py:131-135
    fan_in = w.shape[1] * w.shape[2] * w.shape[3]
    fan_out = w.shape[0] * w.shape[2] * w.shape[3]
    a = math.sqrt(3.0) * math.sqrt(2.0 / (fan_in + fan_out))
    return w.uniform_(-a, a, generator=g)


def init_state_dict(depth=50, seed=0, calibrate=True, num_classes=81, stacked_convs=4, norm=True,
                    stage_with_dcn=(False, False, False, False), rescoring=False):
    """Build a float32 state_dict with the reference's parameter names/shapes
     and init rules (resnet.py:479-497, fpn.py:131-135,
    sipmask_head.py:226-239), then apply the calibration overrides.
    stacked_convs / norm: the SSD configs use stacked_convs=2, norm_cfg=None (towers get conv biases instead of
    GN, sipmask_head.py:161-185 `bias=self.norm_cfg is None`; FeatureAlign flag_norm=False, :196)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, k, bias=False, init="kaiming", std=None):
        w = torch.empty(co, ci, k, k)
        if init == "kaiming":
            _kaiming(w, g)
        elif init == "xavier":
            _xavier_uniform(w, g)
        else:
            w.normal_(0, std, generator=g)
        sd[name + ".weight"] = w
        if bias:
            sd[name + ".bias"] = torch.zeros(co)

    def bn(name, c, gamma=1.0):
        sd[name + ".weight"] = torch.full((c,), gamma)
        sd[name + ".bias"] = torch.zeros(c)
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    # ---- backbone
    conv("backbone.conv1", 64, 3, 7)
    bn("backbone.bn1", 64)
    inplanes = 64
    for li, nblocks in enumerate(ARCH[depth]):
        planes = 64 * 2 ** li
        for bi in range(nblocks):
            p = "backbone.layer%d.%d" % (li + 1, bi)
            conv(p + ".conv1", planes, inplanes, 1)
            bn(p + ".bn1", planes)
            conv(p + ".conv2", planes, planes, 3)
            if stage_with_dcn[li] and bi % 3 == 0:      # SipMask++: DCN in block 0 and every 3rd (resnet.py:270-291)
                # conv_offset is zero-initialised (deform_conv.py:289-291); the calibration gives it small
                # non-zero values so the sampling really moves
                conv(p + ".conv2.conv_offset", 18, planes, 3, bias=True, init="normal", std=0.02 if calibrate else 0.0)
                if not calibrate:
                    sd[p + ".conv2.conv_offset.weight"].zero_()
            bn(p + ".bn2", planes)
            conv(p + ".conv3", planes * 4, planes, 1)
            # zero_init_residual sets bn3.weight = 0 (resnet.py:492-495); the
            # calibration override sets it back to 1 so residual branches are live
            bn(p + ".bn3", planes * 4, gamma=1.0 if calibrate else 0.0)
            if bi == 0:
                conv(p + ".downsample.0", planes * 4, inplanes, 1)
                bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    # ---- neck (start_level=1, 3 laterals, 3 outputs, 2 extra)
    for i, ci in enumerate((512, 1024, 2048)):
        conv("neck.lateral_convs.%d.conv" % i, 256, ci, 1, bias=True, init="xavier")
    for i in range(5):
        conv("neck.fpn_convs.%d.conv" % i, 256, 256, 3, bias=True, init="xavier")
    # ---- head
    h = "bbox_head."
    for kind, n in (("cls", stacked_convs - 1), ("reg", stacked_convs)):
        for i in range(n):
            conv(h + "%s_convs.%d.conv" % (kind, i), 256, 256, 3, bias=not norm, init="normal", std=0.01)
            if norm:
                sd[h + "%s_convs.%d.gn.weight" % (kind, i)] = torch.ones(256)
                sd[h + "%s_convs.%d.gn.bias" % (kind, i)] = torch.zeros(256)
            elif calibrate:        # exercise the bias path of the norm-free towers
                sd[h + "%s_convs.%d.conv.bias" % (kind, i)].normal_(0, 0.05, generator=g)
    ncls = num_classes - 1
    conv(h + "fcos_cls", ncls, 256, 3, bias=True, init="normal", std=0.01)
    sd[h + "fcos_cls.bias"].fill_(float(-math.log((1 - 0.01) / 0.01)))
    conv(h + "fcos_reg", 4, 256, 3, bias=True, init="normal", std=0.01)
    conv(h + "fcos_centerness", 1, 256, 3, bias=True, init="normal", std=0.01)
    for i in range(5):
        sd[h + "scales.%d.scale" % i] = torch.tensor(1.0)
    conv(h + "feat_align.conv_offset", 72, 4, 1, init="normal", std=0.2 if calibrate else 0.0)
    if not calibrate:
        sd[h + "feat_align.conv_offset.weight"].zero_()
    conv(h + "feat_align.conv_adaption", 256, 256, 3, init="normal", std=0.01)
    sd[h + "feat_align.norm.weight"] = torch.ones(256)
    sd[h + "feat_align.norm.bias"] = torch.zeros(256)
    conv(h + "sip_cof", 128, 256, 3, bias=True, init="normal", std=0.05 if calibrate else 0.001)
    conv(h + "sip_mask_lat", 32, 512, 3, bias=True, init="normal", std=0.01)
    conv(h + "sip_mask_lat0", 512, 768, 1, bias=True, init="normal", std=0.01)
    if rescoring:                                   # SipMask++ mask scoring branch, sipmask_head.py:200-219
        chans = [1, 16, 16, 16, 32, 64, 128]
        for i in range(6):
            conv(h + "convs_scoring.%d.conv" % i, chans[i + 1], chans[i], 3, bias=True, init="kaiming")
        conv(h + "mask_scoring", num_classes - 1, 128, 1, bias=True, init="normal", std=0.001 if not calibrate else 0.05)
        if calibrate:
            sd[h + "mask_scoring.bias"].normal_(0.3, 0.1, generator=g)
    if calibrate:
        # random-normal tower weights at std 0.01 shrink the signal by ~0.5x per
        # layer; give the synthetic net O(1) activations so boxes/masks are non-trivial
        for i in range(stacked_convs - 1):
            sd[h + "cls_convs.%d.conv.weight" % i].mul_(3.0 if norm else 1.5)
        for i in range(stacked_convs):
            sd[h + "reg_convs.%d.conv.weight" % i].mul_(3.0 if norm else 1.5)
        sd[h + "feat_align.conv_adaption.weight"].mul_(3.0)
        sd[h + "fcos_reg.weight"].mul_(3.0)
        sd[h + "fcos_reg.bias"].fill_(2.0)
        sd[h + "fcos_cls.weight"].mul_(8.0)
        sd[h + "sip_mask_lat.weight"].mul_(4.0)
        sd[h + "sip_mask_lat0.weight"].mul_(4.0)
        for i in range(5):
            sd[h + "scales.%d.scale" % i] = torch.tensor(1.0 + 0.25 * i)
    return sd


def calibrate_cls_bias(sd, cls_logits_nobias, target=1000, score_thr=0.05):
    """pick fcos_cls.bias by bisection so that about ``target``
    class scores of one image exceed score_thr.  cls_logits_nobias: 1-D tensor of
    all class logits of one image computed with bias 0."""
    lo, hi = -20.0, 20.0
    logit_thr = math.log(score_thr / (1 - score_thr))
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        n = int((cls_logits_nobias + mid > logit_thr).sum())
        if n > target:
            hi = mid
        else:
            lo = mid
    sd["bbox_head.fcos_cls.bias"].fill_(0.5 * (lo + hi))
    return 0.5 * (lo + hi)


# ----------------------------------------------------------------------------
# forward graph
# ----------------------------------------------------------------------------


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def backbone_forward(sd, x, depth=50, prefix="backbone."):
    """resnet.py:501-512; caffe style => stride on conv1 of the block (:125-130)."""
    x = F.conv2d(x, sd[prefix + "conv1.weight"], None, 2, 3)
    x = F.relu(_bn(x, sd, prefix + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, nblocks in enumerate(ARCH[depth]):
        for bi in range(nblocks):
            p = "%slayer%d.%d" % (prefix, li + 1, bi)
            stride = 2 if (bi == 0 and li > 0) else 1
            idt = x
            o = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride), sd, p + ".bn1"))
            if (p + ".conv2.conv_offset.weight") in sd:     # DeformConvPack, M/mmdet/ops/dcn/deform_conv.py:293-296
                w_off = sd[p + ".conv2.conv_offset.weight"]
                off = F.conv2d(o, w_off, sd[p + ".conv2.conv_offset.bias"], 1, 1)
                o = ops.deform_conv(o, off, sd[p + ".conv2.weight"], 1, 1, 1, w_off.shape[0] // 18)
            else:
                o = F.conv2d(o, sd[p + ".conv2.weight"], None, 1, 1)
            o = F.relu(_bn(o, sd, p + ".bn2"))
            o = _bn(F.conv2d(o, sd[p + ".conv3.weight"]), sd, p + ".bn3")
            if bi == 0:
                idt = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), sd, p + ".downsample.1")
            x = F.relu(o + idt)
        outs.append(x)
    return outs


def fpn_forward(sd, feats, prefix="neck."):
    """fpn.py:137-178 with start_level=1, add_extra_convs on outputs, relu before P7."""
    ins = feats[1:]
    lats = [F.conv2d(ins[i], sd[prefix + "lateral_convs.%d.conv.weight" % i],
                     sd[prefix + "lateral_convs.%d.conv.bias" % i]) for i in range(3)]
    for i in (2, 1):
        lats[i - 1] = lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[2:], mode="nearest")
    outs = [F.conv2d(lats[i], sd[prefix + "fpn_convs.%d.conv.weight" % i],
                     sd[prefix + "fpn_convs.%d.conv.bias" % i], 1, 1) for i in range(3)]
    outs.append(F.conv2d(outs[-1], sd[prefix + "fpn_convs.3.conv.weight"],
                         sd[prefix + "fpn_convs.3.conv.bias"], 2, 1))
    outs.append(F.conv2d(F.relu(outs[-1]), sd[prefix + "fpn_convs.4.conv.weight"],
                         sd[prefix + "fpn_convs.4.conv.bias"], 2, 1))
    return outs


def _tower(sd, x, p):
    """ConvModule conv -> (GN32) -> ReLU, conv_module.py:34-132; bias iff there is no norm."""
    x = F.conv2d(x, sd[p + ".conv.weight"], sd.get(p + ".conv.bias"), 1, 1)
    if p + ".gn.weight" in sd:
        x = F.group_norm(x, 32, sd[p + ".gn.weight"], sd[p + ".gn.bias"], 1e-5)
    return F.relu(x)


def tower_depths(sd, prefix="bbox_head."):
    """(#cls convs, #reg convs, norm?) as laid down by _init_layers (sipmask_head.py:159-185)."""
    n = lambda kind: sum(1 for k in sd if k.startswith(prefix + kind + "_convs.") and k.endswith(".conv.weight"))
    return n("cls"), n("reg"), (prefix + "reg_convs.0.gn.weight") in sd


def head_forward(sd, feats, prefix="bbox_head.", strides=FPN_STRIDES, return_aux=False):
    """sipmask_head.py:241-287."""
    h = prefix
    cls_scores, bbox_preds, ctrs, cofs, fm = [], [], [], [], []
    ncls_convs, nreg_convs, flag_norm = tower_depths(sd, h)
    aux = dict(bbox_raw=[], offsets=[], cls_feat=[], reg_feat=[], aligned=[])
    for li, (x, stride) in enumerate(zip(feats, strides)):
        cf, rf = x, x
        for i in range(ncls_convs):
            cf = _tower(sd, cf, h + "cls_convs.%d" % i)
        for i in range(nreg_convs):
            rf = _tower(sd, rf, h + "reg_convs.%d" % i)
        bbox_pred = sd[h + "scales.%d.scale" % li] * F.conv2d(rf, sd[h + "fcos_reg.weight"], sd[h + "fcos_reg.bias"], 1, 1)
        # FeatureAlign.forward: offset = self.conv_offset(shape.detach()) (sipmask_head.py:50) -- no gradient
        # flows from the deformable conv's offsets back into the box branch
        offset = F.conv2d(bbox_pred.detach(), sd[h + "feat_align.conv_offset.weight"])
        y = ops.deform_conv(cf, offset, sd[h + "feat_align.conv_adaption.weight"], 1, 1, 1, 4)
        if flag_norm:                             # FeatureAlign.forward, sipmask_head.py:49-55
            y = F.group_norm(y, 32, sd[h + "feat_align.norm.weight"], sd[h + "feat_align.norm.bias"], 1e-5)
        y = F.relu(y)
        cls_scores.append(F.conv2d(y, sd[h + "fcos_cls.weight"], sd[h + "fcos_cls.bias"], 1, 1))
        ctrs.append(F.conv2d(rf, sd[h + "fcos_centerness.weight"], sd[h + "fcos_centerness.bias"], 1, 1))
        bbox_preds.append(bbox_pred.float() * stride)
        cofs.append(F.conv2d(y, sd[h + "sip_cof.weight"], sd[h + "sip_cof.bias"], 1, 1))
        if li < 3:
            fm.append(rf if li == 0 else F.interpolate(rf, scale_factor=2 ** li, mode="bilinear", align_corners=False))
        if return_aux:
            aux["bbox_raw"].append(bbox_pred)
            aux["offsets"].append(offset)
            aux["cls_feat"].append(cf)
            aux["reg_feat"].append(rf)
            aux["aligned"].append(y)
    fmc = torch.cat(fm, 1)
    lat0 = F.relu(F.conv2d(fmc, sd[h + "sip_mask_lat0.weight"], sd[h + "sip_mask_lat0.bias"]))
    lat = F.relu(F.conv2d(lat0, sd[h + "sip_mask_lat.weight"], sd[h + "sip_mask_lat.bias"], 1, 1))
    feat_masks = F.interpolate(lat, scale_factor=4, mode="bilinear", align_corners=False)
    if return_aux:
        aux["basis_lowres"] = lat
        return cls_scores, bbox_preds, ctrs, cofs, feat_masks, aux
    return cls_scores, bbox_preds, ctrs, cofs, feat_masks


def get_points(featmap_sizes, strides=FPN_STRIDES):
    """sipmask_head.py:664-695: (x*s + s//2, y*s + s//2), row-major."""
    pts = []
    for (h, w), s in zip(featmap_sizes, strides):
        xr = torch.arange(0, w * s, s, dtype=torch.float32)
        yr = torch.arange(0, h * s, s, dtype=torch.float32)
        y, x = torch.meshgrid(yr, xr, indexing="ij")
        pts.append(torch.stack((x.reshape(-1), y.reshape(-1)), -1) + s // 2)
    return pts


def select_candidates(cls_scores, bbox_preds, ctrs, cofs, img_shape, nms_pre=1000,
                      strides=FPN_STRIDES):
    """sipmask_head.py:563-591 for ONE image (tensors [C,h,w] per level).
    Returns mlvl_bboxes [K,4], mlvl_scores [K,C+1], mlvl_centerness [K],
    mlvl_cofs [K,128], and (level, position) of every candidate."""
    pts = get_points([c.shape[-2:] for c in cls_scores], strides)
    B_, S_, C_, F_, lv, ps = [], [], [], [], [], []
    for li, (cs, bp, ct, cf, p) in enumerate(zip(cls_scores, bbox_preds, ctrs, cofs, pts)):
        ncls = cs.shape[0]
        scores = ops.sigmoid_ref(cs.permute(1, 2, 0).reshape(-1, ncls))      # see ops.sigmoid_ref: f64, rounded once
        ctr = ops.sigmoid_ref(ct.permute(1, 2, 0).reshape(-1))
        bp = bp.permute(1, 2, 0).reshape(-1, 4)
        cf = cf.permute(1, 2, 0).reshape(-1, 128)
        pos = torch.arange(scores.shape[0])
        if nms_pre > 0 and scores.shape[0] > nms_pre:
            mx = (scores * ctr[:, None]).max(dim=1)[0]
            idx = torch.from_numpy(ops.topk_desc(mx.numpy(), nms_pre))
            p, bp, cf, scores, ctr, pos = p[idx], bp[idx], cf[idx], scores[idx], ctr[idx], pos[idx]
        B_.append(ops.distance2bbox(p, bp, max_shape=img_shape))
        S_.append(scores)
        C_.append(ctr)
        F_.append(cf)
        lv.append(torch.full((scores.shape[0],), li, dtype=torch.long))
        ps.append(pos)
    scores = torch.cat(S_)
    scores = torch.cat([scores.new_zeros(scores.shape[0], 1), scores], 1)
    return torch.cat(B_), scores, torch.cat(C_), torch.cat(F_), torch.cat(lv), torch.cat(ps)


def get_masks_single(cls_scores, bbox_preds, ctrs, cofs, feat_mask, img_shape, cfg,
                     scale_factor=1.0, rescale=False, ssd_flag=False):
    """get_bboxes_single up to (not including) RLE: sipmask_head.py:543-633.
    scale_factor: scalar, or the [w,h,w,h] array of a keep_ratio=False pipeline (ssd_flag configs)."""
    mb, ms, mc, mf, lv, ps = select_candidates(cls_scores, bbox_preds, ctrs, cofs, img_shape,
                                               cfg.get("nms_pre", -1))
    if rescale:
        mb = mb / torch.as_tensor(scale_factor, dtype=torch.float32)            # :587-588
    out = dict(cand_boxes=mb, cand_scores=ms, cand_ctr=mc, cand_cofs=mf, cand_level=lv, cand_pos=ps)
    if not ssd_flag:
        det, lab, keep = ops.multiclass_nms_idx(mb.numpy(), ms.numpy(), cfg["score_thr"],
                                                cfg["nms"]["iou_thr"], cfg["max_per_img"],
                                                score_factors=mc.numpy())
        det_cofs = mf[torch.from_numpy(keep)]
        out.update(det_bboxes=det, det_labels=lab, idxs_keep=keep)
    else:                                                                          # :603-605
        sc = (ms * mc.view(-1, 1))[:, 1:].t().contiguous()
        det, lab, det_cofs = ops.fast_nms(mb.numpy(), sc.numpy(), mf.numpy(), cfg["nms"]["iou_thr"], 200,
                                          cfg["score_thr"])
        det_cofs = torch.from_numpy(det_cofs)
        out.update(det_bboxes=det, det_labels=lab)
    if det.shape[0] > 0:
        out.update(ops.mask_assemble(feat_mask, det_cofs, det, scale_factor, rescale, ssd_flag=ssd_flag))
        out["det_cofs"] = det_cofs
    return out


def mask_rescoring(sd, pos_masks, det_labels, det_scores, prefix="bbox_head."):
    """SipMask++ rescoring, sipmask_head.py:635-641: six ConvModule(3x3, stride 2, no padding, bias, ReLU) on each
    cropped probability mask, 1x1 mask_scoring + ReLU, global max pool, the detection's class channel, times the
    box score.  pos_masks [N,Hm,Wm] (CropSplit output), det_labels [N] long, det_scores [N] -> [N]."""
    h = prefix
    x = torch.as_tensor(pos_masks, dtype=torch.float32).unsqueeze(1)
    for i in range(6):
        x = F.relu(F.conv2d(x, sd[h + "convs_scoring.%d.conv.weight" % i], sd[h + "convs_scoring.%d.conv.bias" % i], 2, 0))
    x = F.relu(F.conv2d(x, sd[h + "mask_scoring.weight"], sd[h + "mask_scoring.bias"]))
    x = F.max_pool2d(x, kernel_size=x.shape[2:]).squeeze(-1).squeeze(-1)
    lab = torch.as_tensor(det_labels, dtype=torch.long)
    return x[torch.arange(x.shape[0]), lab] * torch.as_tensor(det_scores, dtype=torch.float32)


def detector_forward(sd, img, depth=50):
    feats = backbone_forward(sd, img, depth)
    pyr = fpn_forward(sd, feats)
    return head_forward(sd, pyr)


DEFAULT_TEST_CFG = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05,
                        nms=dict(type="nms", iou_thr=0.5), max_per_img=100)
