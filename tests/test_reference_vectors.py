"""The oracle against outputs of the REFERENCE's own Python code (tests/golden/ref_vectors.npz).

The fixtures were produced in the build container by tests/golden/make_reference_vectors.py, which executes the
reference's source files (SipMaskHead.forward / get_bboxes / loss / fcos_target / fast_nms, mmdet's distance2bbox,
bbox_overlaps, multiclass_nms_idx, the loss modules) with stand-ins only for the compiled extensions and absent
third-party packages (listed in the fixture's `meta`).  Inputs are re-created exactly by oracle/fixtures.py, so these
tests need neither the reference nor a GPU.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as FX
from oracle import loss as OL
from oracle import model as OM
from oracle import ops as O

HERE = os.path.dirname(os.path.abspath(__file__))
NUM_CLASSES = 9
CFG = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type="nms", iou_thr=0.5), max_per_img=100)


@pytest.fixture(scope="module")
def ref():
    z = np.load(os.path.join(HERE, "golden", "ref_vectors.npz"))
    meta = json.loads(str(z["meta"]))
    assert meta["num_classes"] == NUM_CLASSES and "mmdet.ops.DeformConv" in meta["stand_ins"]
    return z


def _head_sd(stacked_convs=4, norm=True):
    tmpl = {k: v for k, v in OM.init_state_dict(50, 0, num_classes=NUM_CLASSES, stacked_convs=stacked_convs, norm=norm).items()
            if k.startswith("bbox_head.")}
    sd = FX.head_state_dict({k[len("bbox_head."):]: v for k, v in tmpl.items()})
    return {"bbox_head." + k: v for k, v in sd.items()}


def _check_summary(ref, name, t):
    a = t.detach().numpy().astype(np.float32).reshape(-1)
    assert tuple(ref[name + ".shape"]) == tuple(t.shape), name
    idx = np.unique(np.linspace(0, a.size - 1, 48).astype(np.int64))
    scale = float(ref[name + ".abssum"]) / a.size + 1e-12
    np.testing.assert_allclose(a[idx], ref[name + ".samples"], rtol=1e-4, atol=1e-4 * scale, err_msg=name)
    np.testing.assert_allclose(np.abs(a.astype(np.float64)).sum(), float(ref[name + ".abssum"]), rtol=1e-5, err_msg=name)
    np.testing.assert_allclose(a.astype(np.float64).sum(), float(ref[name + ".sum"]), rtol=1e-4,
                               atol=1e-6 * float(ref[name + ".abssum"]), err_msg=name)


@pytest.mark.parametrize("tag,kw", [("A_forward_gn", dict()), ("A_forward_ssd", dict(stacked_convs=2, norm=False))])
def test_head_forward_matches_reference_forward(ref, tag, kw):
    """SipMaskHead.forward, sipmask_head.py:241-287 (towers, Scale, FeatureAlign, predictors, mask basis)"""
    sd = _head_sd(**kw)
    feats = FX.pyramid_feats(11, 2)
    with torch.no_grad():
        cls, box, ctr, cof, fm = OM.head_forward(sd, feats)
    for l in range(5):
        _check_summary(ref, "%s.cls%d" % (tag, l), cls[l])
        _check_summary(ref, "%s.box%d" % (tag, l), box[l])
        _check_summary(ref, "%s.ctr%d" % (tag, l), ctr[l])
        _check_summary(ref, "%s.cof%d" % (tag, l), cof[l])
    _check_summary(ref, tag + ".feat_mask", fm)


@pytest.mark.parametrize("tag,rescale,sf,ssd", [("B_default", None, 1.0, False), ("B_rescale", True, 1.5, False),
                                                ("B_ssd", True, np.array([1.25, 1.5, 1.25, 1.5], np.float32), True)])
def test_postprocessing_matches_reference_get_bboxes(ref, tag, rescale, sf, ssd):
    """get_bboxes_single, sipmask_head.py:543-663: candidate selection, multiclass_nms_idx / fast_nms, mask assembly,
    paste into the (original) image canvas.  Boxes to 1e-5, labels and EVERY mask pixel exact."""
    H, W = FX.IMG_HW
    cls, box, ctr, cof, fm = FX.head_outputs(21, 2, NUM_CLASSES - 1)
    mh, mw = [int(v) for v in ref[tag + ".mask_hw"]]
    ndet = 0
    for b in range(2):
        r = OM.get_masks_single([c[b] for c in cls], [x[b] for x in box], [c[b] for c in ctr], [c[b] for c in cof], fm[b],
                                (H, W, 3), CFG, scale_factor=sf, rescale=rescale, ssd_flag=ssd)
        det, lab = ref["%s.det%d" % (tag, b)], ref["%s.lab%d" % (tag, b)]
        assert r["det_bboxes"].shape == det.shape
        np.testing.assert_array_equal(r["det_labels"], lab)
        np.testing.assert_allclose(r["det_bboxes"], det, rtol=1e-5, atol=1e-5)
        want = np.unpackbits(ref["%s.masks%d" % (tag, b)], axis=1)[:, :mh * mw].reshape(-1, mh, mw)
        got = np.zeros_like(want)
        m = r["masks"].numpy()
        hh, ww = min(mh, m.shape[1]), min(mw, m.shape[2])
        got[:, :hh, :ww] = m[:, :hh, :ww]                      # the paste of sipmask_head.py:645-653
        assert int((got != want).sum()) == 0, (tag, b, int((got != want).sum()))
        ndet += det.shape[0]
    assert ndet > 50


@pytest.mark.parametrize("tag,cs", [("C_loss_cs", True), ("C_loss_nocs", False)])
def test_loss_and_targets_match_reference(ref, tag, cs):
    """SipMaskHead.loss (:290-498), fcos_target / fcos_target_single (:731-857), centerness_target (:859-866)"""
    cls, box, ctr, cof, fm = FX.head_outputs(31, 2, NUM_CLASSES - 1)
    cof = [c * 0.25 for c in cof]
    gtb, gtl, gtm = FX.ground_truth(32, 2, NUM_CLASSES - 1)
    losses, aux = OL.head_loss(cls, box, ctr, cof, fm * 0.25, gtb, gtl, gtm, center_sampling=cs)
    np.testing.assert_array_equal(aux["labels"].numpy(), ref[tag + ".labels"])
    np.testing.assert_array_equal(aux["bbox_targets"].numpy(), ref[tag + ".bbox_targets"])
    for b in range(2):
        np.testing.assert_array_equal(aux["per_img"][b][2].numpy(), ref["%s.gt_inds%d" % (tag, b)])
    pts = torch.cat(OM.get_points([c.shape[-2:] for c in cls])).numpy()
    np.testing.assert_array_equal(pts, ref[tag + ".points"])
    assert int((ref[tag + ".labels"] > 0).sum()) > 10
    for k in ("loss_cls", "loss_bbox", "loss_centerness", "loss_mask"):
        np.testing.assert_allclose(float(losses[k]), float(ref["%s.%s" % (tag, k)]), rtol=2e-5, err_msg=k)


def _boxes(s1, s2, n, grow=0.0):
    a = torch.from_numpy(np.concatenate([FX.exact(s1, (n, 2), 0, 2 ** 11, 2.0 ** -4), FX.exact(s2, (n, 2), 0, 2 ** 11, 2.0 ** -4)], 1))
    return torch.cat([torch.min(a[:, :2], a[:, 2:]), torch.max(a[:, :2], a[:, 2:]) + grow], 1)


def test_small_functions_match_reference(ref):
    a, b = _boxes(41, 42, 40), _boxes(43, 44, 40)
    np.testing.assert_allclose(np.asarray(O.bbox_overlaps(a, b)), ref["D_overlaps.full"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.asarray(O.bbox_overlaps(a, b, is_aligned=True)), ref["D_overlaps.aligned"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(OL._aligned_iou(a, b).numpy(), ref["D_overlaps.aligned"], rtol=1e-6, atol=1e-7)
    pts = torch.from_numpy(FX.exact(45, (64, 2), 0, 2 ** 11, 2.0 ** -4))
    dist = torch.from_numpy(FX.exact(46, (64, 4), -64, 2 ** 11, 2.0 ** -4))
    np.testing.assert_array_equal(np.asarray(O.distance2bbox(pts, dist)), ref["D_distance2bbox.plain"])
    np.testing.assert_array_equal(np.asarray(O.distance2bbox(pts, dist, max_shape=(96, 128, 3))), ref["D_distance2bbox.clamped"])
    np.testing.assert_array_equal(OL._d2b(pts, dist).numpy(), ref["D_distance2bbox.plain"])
    t = torch.from_numpy(FX.exact(47, (50, 4), 1, 2 ** 10, 2.0 ** -4))
    np.testing.assert_allclose(OL.centerness_target(t).numpy(), ref["D_centerness_target"], rtol=1e-6)


def test_cuda_semantics_crop_split_vs_reference_python_crop_split(ref):
    """The reference keeps a python crop_split (sipmask_head.py:58-105) next to the CUDA op it actually calls; the oracle
    restates the CUDA kernel.  The two rules (python: pixel < (x1+x2)/2; CUDA: int((x - x1) / ((x2 - x1 + 0.1) / 2)))
    can only disagree on the one pixel row / column next to a box centre, so on 12 quarter-pixel boxes the restated
    kernel must reproduce the reference's python result except for a handful of such pixels."""
    data = FX.exact(48, (4, 24, 32, 12), 0, 2 ** 10, 2.0 ** -10)
    rois = torch.from_numpy(FX.exact(49, (12, 4), 0, 96, 0.25))
    rois = torch.cat([torch.min(rois[:, :2], rois[:, 2:]), torch.max(rois[:, :2], rois[:, 2:]) + 2.0], 1).numpy()
    got = O.crop_split(data, rois, 2)
    want = ref["D_py_crop_split"]
    assert (want != 0).sum() > 900
    diff = (got != want)
    assert int(diff.sum()) <= 8 and int((diff.sum((0, 1)) == 0).sum()) >= 10, diff.sum((0, 1))
    for n in np.nonzero(diff.sum((0, 1)))[0]:                       # only next to the centre lines of that box
        ys, xs = np.nonzero(diff[:, :, n])
        cx, cy = (rois[n, 0] + rois[n, 2]) / 2, (rois[n, 1] + rois[n, 3]) / 2
        assert all(abs(x - cx) <= 1 or abs(y - cy) <= 1 for x, y in zip(xs, ys))


def test_fast_nms_matches_reference(ref):
    """SipMaskHead.fast_nms + jaccard + intersect, sipmask_head.py:868-960"""
    boxes = _boxes(51, 52, 300, grow=1.0).numpy()
    scores = FX.exact_unique(53, (8, 300))
    cofs = FX.exact(54, (300, 128))
    det, lab, m = O.fast_nms(boxes, scores, cofs, 0.5, 200, 0.6, max_out=100)
    np.testing.assert_array_equal(lab, ref["E_fast_nms.lab"])
    np.testing.assert_allclose(det, ref["E_fast_nms.det"], rtol=1e-6)
    np.testing.assert_allclose(m.astype(np.float64).sum(1), ref["E_fast_nms.cof_rowsum"], rtol=1e-6)
    assert det.shape[0] > 20


# ---------------------------------------------------------------------------------------------------------------
# SipMask-VIS head (V/mmdet/models/anchor_heads/sipmask_head.py)
# ---------------------------------------------------------------------------------------------------------------
VIS_CLASSES = 5
VIS_CFG = dict(nms_pre=200, min_bbox_size=0, score_thr=0.1, nms=dict(type="nms", iou_thr=0.5), max_per_img=10)


def test_vis_forward_matches_reference(ref):
    """forward(flag_train=False), V/...:252-317: the M/ head with 3-deep towers plus track_convs -> sipmask_track"""
    from oracle import vis as OV
    tmpl = {k[len("bbox_head."):]: v for k, v in OV.init_vis_state_dict(0, num_classes=VIS_CLASSES).items()
            if k.startswith("bbox_head.")}
    sd = {"bbox_head." + k: v for k, v in FX.head_state_dict(tmpl, seed=300).items()}
    feats = FX.pyramid_feats(61, 1)
    with torch.no_grad():
        cls, box, ctr, cof, fm = OM.head_forward(sd, feats)
        tf = OV.track_forward(sd, feats)
    for l in range(5):
        _check_summary(ref, "F_vis_forward.cls%d" % l, cls[l])
        _check_summary(ref, "F_vis_forward.box%d" % l, box[l])
        _check_summary(ref, "F_vis_forward.cof%d" % l, cof[l])
    _check_summary(ref, "F_vis_forward.feat_mask", fm)
    _check_summary(ref, "F_vis_forward.track", tf)


@pytest.mark.parametrize("case,rescale,sf", [("G_vis_clip", False, 1.0), ("G_vis_clip_rescale", True, 1.5)])
def test_vis_clip_matches_reference_get_bboxes(ref, case, rescale, sf):
    """get_bboxes over a 4-frame clip, V/...:565-684: fast_nms detections, centre features, comprehensive matching
    scores, identity assignment and the update of the tracker memory; masks pasted per object id."""
    from oracle import vis as OV
    H, W = FX.IMG_HW
    mh, mw = [int(v) for v in ref[case + ".mask_hw"]]
    trk = OV.Tracker()
    total = 0
    for f in range(4):
        cls, box, ctr, cof, fm = FX.head_outputs(71 + (f // 2), 1, VIS_CLASSES - 1)
        box = [b + 0.5 * f for b in box]
        tf = FX.texact(81 + f, (1, 512, H // 8, W // 8), -2 ** 9, 2 ** 9, 2.0 ** -10)
        r = OV.get_masks_single_vis([c[0] for c in cls], [x[0] for x in box], [c[0] for c in ctr], [c[0] for c in cof], fm[0],
                                    (H, W, 3), VIS_CFG, scale_factor=sf, rescale=rescale)
        tag = "%s.f%d" % (case, f)
        det = r["det_bboxes"]
        np.testing.assert_array_equal(r["det_labels"], ref[tag + ".lab"])
        np.testing.assert_allclose(det, ref[tag + ".det"], rtol=1e-5, atol=1e-5)
        boxes = torch.from_numpy(det[:, :4].copy()) * (sf if rescale else 1.0)
        feat = OV.extract_box_feature_center(tf[0], boxes)
        ids = trk.step(det, r["det_labels"], feat, is_first=(f == 0))
        np.testing.assert_array_equal(np.asarray(ids, np.int64), ref[tag + ".ids"])
        # the reference keeps, per object id, the mask of the LAST detection carrying it (dict overwrite, :672-680)
        want_ids = ref[tag + ".mask_ids"]
        want = np.unpackbits(ref[tag + ".masks"], axis=1)[:, :mh * mw].reshape(-1, mh, mw)
        m = r["masks"].numpy()
        for j, oid in enumerate(want_ids):
            i = int(np.nonzero(np.asarray(ids) == oid)[0][-1])
            got = np.zeros((mh, mw), np.uint8)
            hh, ww = min(mh, m.shape[1]), min(mw, m.shape[2])
            got[:hh, :ww] = m[i, :hh, :ww]
            assert int((got != want[j]).sum()) == 0, (tag, oid)
        assert sorted(set(int(i) for i in ids if i >= 0)) == [int(v) for v in want_ids]
        total += det.shape[0]
    np.testing.assert_allclose(trk.prev_bboxes.numpy(), ref[case + ".memory_boxes"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(trk.prev_det_labels.numpy(), ref[case + ".memory_labels"])
    assert total >= 20 and int(ref[case + ".f3.ids"].max()) >= 3


# ---------------------------------------------------------------------------------------------------------------
# maskrcnn-benchmark variant (B/fcos_core/modeling/rpn/sipmask)
# ---------------------------------------------------------------------------------------------------------------
def test_benchmark_head_forward_matches_reference(ref):
    """SipMaskHead.forward in eval mode, B/...sipmask.py:145-190 (NORM_REG_TARGETS, CENTERNESS_ON_REG as in the yaml)"""
    from oracle import fcos_core as OB
    tmpl = {k[len("rpn.head."):]: v for k, v in OB.init_head_state_dict(0, num_classes=NUM_CLASSES).items()}
    sd = FX.head_state_dict(tmpl, seed=500)
    sd["bbox_pred.bias"] = sd["bbox_pred.bias"] + 1.0
    sd = {"rpn.head." + k: v for k, v in sd.items()}
    feats = FX.pyramid_feats(91, 2)
    with torch.no_grad():
        logits, reg, ctr, cof, fm = OB.head_forward(sd, feats)
    for l in range(5):
        _check_summary(ref, "H_b_forward.cls%d" % l, logits[l])
        _check_summary(ref, "H_b_forward.box%d" % l, reg[l])
        _check_summary(ref, "H_b_forward.ctr%d" % l, ctr[l])
        _check_summary(ref, "H_b_forward.cof%d" % l, cof[l])
    _check_summary(ref, "H_b_forward.feat_mask", fm)


@pytest.mark.parametrize("tag,ori_wh", [("I_b_post_same", (128, 96)), ("I_b_post_rescale", (85, 64))])
def test_benchmark_postprocessor_matches_reference(ref, tag, ori_wh):
    """SipMaskPostProcessor.forward, B/...inference.py:66-236: (location, class) candidates, sqrt scores, clip, ml_nms,
    kthvalue cut, mask assembly at 2 / scale_factor, paste; and compute_locations (sipmask.py:261-285)"""
    from oracle import fcos_core as OB
    H, W = FX.IMG_HW
    cls, box, ctr, cof, fm = FX.head_outputs(95, 2, NUM_CLASSES - 1)
    pts = torch.cat(OM.get_points([c.shape[-2:] for c in cls])).numpy()
    np.testing.assert_array_equal(pts, ref["I_b_post.locations"])
    mh, mw = [int(v) for v in ref[tag + ".mask_hw"]]
    n = 0
    for b in range(2):
        r = OB.postprocess_single([c[b] for c in cls], [x[b] for x in box], [c[b] for c in ctr], [c[b] for c in cof], fm[b],
                                  (H, W), ori_wh)
        np.testing.assert_array_equal(r["labels"].numpy(), ref["%s.labels%d" % (tag, b)])
        np.testing.assert_allclose(r["bbox"].numpy(), ref["%s.bbox%d" % (tag, b)], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(r["scores"].numpy(), ref["%s.scores%d" % (tag, b)], rtol=1e-6)
        want = np.unpackbits(ref["%s.masks%d" % (tag, b)], axis=1)[:, :mh * mw].reshape(-1, mh, mw)
        assert int((r["mask"].numpy() != want).sum()) == 0
        n += want.shape[0]
    assert n > 50


# ---------------------------------------------------------------------------------------------------------------
# backbone and neck (M/mmdet/models/backbones/resnet.py, M/mmdet/models/necks/fpn.py)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,dcn", [("J_backbone", None), ("J_backbone_dcn", (False, True, True, True))])
def test_backbone_and_fpn_match_reference(ref, tag, dcn):
    """ResNet-50 caffe (frozen BN, eval) plain / with DeformConvPack in stages 2-4, and the SipMask FPN
    (start_level=1, extra convs on outputs, relu before the extra convs)"""
    full = OM.init_state_dict(50, 0, stage_with_dcn=dcn) if dcn else OM.init_state_dict(50, 0)
    tmpl = {k[len("backbone."):]: v for k, v in full.items() if k.startswith("backbone.")}
    sd = {"backbone." + k: v for k, v in FX.trunk_state_dict(tmpl).items()}
    ntmpl = {k[len("neck."):]: v for k, v in full.items() if k.startswith("neck.")}
    sd.update({"neck." + k: v for k, v in FX.trunk_state_dict(ntmpl, seed=900).items()})
    img = FX.texact(601, (1, 3, 64, 96), -2 ** 11, 2 ** 11, 2.0 ** -10)
    with torch.no_grad():
        feats = OM.backbone_forward(sd, img, 50)
        for i, f in enumerate(feats):
            _check_summary(ref, "%s.c%d" % (tag, i + 2), f)
        if dcn is None:
            for i, f in enumerate(OM.fpn_forward(sd, feats)):
                _check_summary(ref, "J_fpn.p%d" % (i + 3), f)


def test_mask_rescoring_matches_reference(ref):
    """SipMask++ rescoring inside get_bboxes_single (sipmask_head.py:635-643) on the SSD-style head: six stride-2
    ConvModules + mask_scoring on every cropped probability mask, class channel, times the box score"""
    tmpl = {k: v for k, v in OM.init_state_dict(50, 0, num_classes=NUM_CLASSES, stacked_convs=2, norm=False,
                                                rescoring=True).items() if k.startswith("bbox_head.")}
    sd = {"bbox_head." + k: v for k, v in FX.head_state_dict({k[len("bbox_head."):]: v for k, v in tmpl.items()}).items()}
    big = [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]
    cls, box, ctr, cof, fm = FX.head_outputs(25, 1, NUM_CLASSES - 1, sizes=big)
    sf = np.array([1.25, 1.5, 1.25, 1.5], np.float32)
    r = OM.get_masks_single([c[0] for c in cls], [x[0] for x in box], [c[0] for c in ctr], [c[0] for c in cof], fm[0],
                            (256, 256, 3), CFG, scale_factor=sf, rescale=True, ssd_flag=True)
    np.testing.assert_array_equal(r["det_labels"], ref["B_rescoring.lab0"])
    np.testing.assert_allclose(r["det_bboxes"], ref["B_rescoring.det0"], rtol=1e-5, atol=1e-5)
    with torch.no_grad():
        ms = OM.mask_rescoring(sd, r["pos_masks"], r["det_labels"], r["det_bboxes"][:, 4])
    want = ref["B_rescoring.mask_scores0"]
    assert want.shape[0] > 20 and float(np.abs(want).max()) > 0
    np.testing.assert_allclose(ms.numpy(), want, rtol=1e-4, atol=1e-6 * float(np.abs(want).max()))


def test_vis_loss_with_matching_term_matches_reference(ref):
    """V/ SipMaskHead.loss (:320-541): the M/ terms plus loss_match over jittered reference-frame boxes (the jitter the
    reference drew from torch's global RNG is part of the fixture)"""
    from oracle import vis as OV
    H, W = FX.IMG_HW
    cls, box, ctr, cof, fm = FX.head_outputs(111, 2, VIS_CLASSES - 1)
    box = [torch.cat([b[:, :2], b[:, :2]], 1) for b in box]
    cof = [c * 0.25 for c in cof]
    gtb, gtl, gtm = FX.ground_truth(112, 2, VIS_CLASSES - 1)
    tf = FX.texact(113, (2, 512, H // 8, W // 8), -2 ** 9, 2 ** 9, 2.0 ** -10)
    tfr = FX.texact(114, (2, 512, H // 8, W // 8), -2 ** 9, 2 ** 9, 2.0 ** -10)
    refb = [b + 2.0 for b in gtb]
    pids = [torch.from_numpy(np.random.RandomState(115 + i).randint(0, len(b) + 1, size=len(b)).astype(np.int64))
            for i, b in enumerate(gtb)]
    jitter = [torch.from_numpy(ref["K_vis_loss.jitter%d" % i]) for i in range(2)]
    losses, aux = OL.head_loss(cls, box, ctr, cof, fm * 0.25, gtb, gtl, gtm, center_sampling=True, stride_norm=False)
    for k in ("loss_cls", "loss_bbox", "loss_centerness", "loss_mask"):
        np.testing.assert_allclose(float(losses[k]), float(ref["K_vis_loss.%s" % k]), rtol=2e-5, err_msg=k)
    lm = OV.track_loss(tf, tfr, aux["mask_aux"], refb, pids, jitter)
    assert float(ref["K_vis_loss.loss_match"]) > 0
    np.testing.assert_allclose(float(lm), float(ref["K_vis_loss.loss_match"]), rtol=2e-5)


def test_loss_with_rescoring_term_matches_reference(ref):
    """SipMask++ training: SipMaskHead.loss with rescoring_flag (sipmask_head.py:463-491) -- loss_iou = weighted squared
    error between the rescoring branch's prediction on the cropped masks and the mask IoU with the (uncropped) gt"""
    tmpl = {k[len("bbox_head."):]: v for k, v in OM.init_state_dict(50, 0, num_classes=NUM_CLASSES, stacked_convs=2, norm=False,
                                                                    rescoring=True).items() if k.startswith("bbox_head.")}
    sd = {"bbox_head." + k: v for k, v in FX.head_state_dict(tmpl).items()}
    big = [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]
    cls, box, ctr, cof, fm = FX.head_outputs(35, 2, NUM_CLASSES - 1, sizes=big)
    cof = [c * 0.25 for c in cof]
    gtb, gtl, gtm = FX.ground_truth(36, 2, NUM_CLASSES - 1, img_hw=(256, 256), max_gt=4)
    with torch.no_grad():
        losses, _ = OL.head_loss(cls, box, ctr, cof, fm * 0.25, gtb, gtl, gtm, rescoring_sd=sd)
    assert float(ref["C_loss_rescoring.loss_iou"]) > 0
    for k in ("loss_cls", "loss_bbox", "loss_centerness", "loss_mask", "loss_iou"):
        np.testing.assert_allclose(float(losses[k]), float(ref["C_loss_rescoring.%s" % k]), rtol=5e-5, err_msg=k)


def test_benchmark_loss_matches_reference(ref):
    """SipMaskLossComputation.__call__, B/...loss.py:330-487 (targets with center sampling, focal / GIoU / centerness
    terms normalised as in the yaml config, mask loss with the 0.9 NMS filter and the halving above 1.0)"""
    from oracle import fcos_core as OB
    cls, box, ctr, cof, fm = FX.head_outputs(97, 2, NUM_CLASSES - 1)
    box = [b / s for b, s in zip(box, FX.STRIDES)]
    cof = [c * 0.25 for c in cof]
    gtb, gtl, gtm = FX.ground_truth(98, 2, NUM_CLASSES - 1)
    losses, labels = OB.loss(cls, box, ctr, cof, fm * 0.25, gtb, gtl, gtm)
    assert int((labels > 0).sum()) > 10
    for k in ("loss_cls", "loss_reg", "loss_centerness", "loss_mask"):
        np.testing.assert_allclose(float(losses[k]), float(ref["L_b_loss.%s" % k]), rtol=2e-5, err_msg=k)


@pytest.mark.parametrize("tag,cs", [("C_loss_cs", True), ("C_loss_nocs", False)])
def test_product_target_assignment_matches_reference(ref, tag, cs):
    """the PRODUCT's host-side target code (sipmask_amd/targets.py, pure torch, runs without a GPU) against the
    reference's fcos_target / get_points / centerness_target / distance2bbox outputs directly"""
    from sipmask_amd import targets as T
    cls = FX.head_outputs(31, 2, NUM_CLASSES - 1)[0]
    gtb, gtl, _ = FX.ground_truth(32, 2, NUM_CLASSES - 1)
    sizes = [tuple(c.shape[-2:]) for c in cls]
    pts = T.level_points(sizes, FX.STRIDES)
    np.testing.assert_array_equal(torch.cat(pts).numpy(), ref[tag + ".points"])
    lab_lvl, tgt_lvl, lab_img, tgt_img, gt_inds = T.fcos_target(pts, FX.STRIDES, OL.REGRESS_RANGES, gtb, gtl, cs, 1.5)
    np.testing.assert_array_equal(torch.cat(lab_lvl).numpy(), ref[tag + ".labels"])
    np.testing.assert_array_equal(torch.cat(tgt_lvl).numpy(), ref[tag + ".bbox_targets"])
    for b in range(2):
        np.testing.assert_array_equal(gt_inds[b].numpy(), ref["%s.gt_inds%d" % (tag, b)])
    t = torch.from_numpy(FX.exact(47, (50, 4), 1, 2 ** 10, 2.0 ** -4))
    np.testing.assert_allclose(T.centerness_target(t).numpy(), ref["D_centerness_target"], rtol=1e-6)
    p = torch.from_numpy(FX.exact(45, (64, 2), 0, 2 ** 11, 2.0 ** -4))
    d = torch.from_numpy(FX.exact(46, (64, 4), -64, 2 ** 11, 2.0 ** -4))
    np.testing.assert_array_equal(T.distance2bbox(p, d).numpy(), ref["D_distance2bbox.plain"])
    np.testing.assert_array_equal(T.distance2bbox(p, d, max_shape=(96, 128, 3)).numpy(), ref["D_distance2bbox.clamped"])
    a, b = _boxes(41, 42, 40), _boxes(43, 44, 40)
    np.testing.assert_allclose(T.aligned_iou(a, b).numpy(), ref["D_overlaps.aligned"], rtol=1e-6, atol=1e-7)
