"""CPU tests of the training-side host logic and of the oracle's own restatements (no GPU needed):
FCOS target assignment (sipmask_amd.targets) == oracle == a brute-force per-point loop; the oracle's
deformable-conv backward == autograd of its forward; the oracle mask loss == the numpy CropSplit ops."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import loss as OL
from oracle import ops as O
from oracle.model import get_points
from sipmask_amd import targets as T

SIZES = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
STRIDES = (8, 16, 32, 64, 128)


def _gt(seed, n):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(n, 2, generator=g) * torch.tensor([100.0, 80.0])
    wh = torch.rand(n, 2, generator=g) * 80 + 6
    return torch.cat([xy, xy + wh], 1), torch.randint(1, 81, (n,), generator=g)


def test_fcos_target_matches_oracle_and_bruteforce():
    pts = get_points(SIZES, STRIDES)
    pts2 = T.level_points(SIZES, STRIDES)
    assert all(torch.equal(a, b) for a, b in zip(pts, pts2))
    nums = [p.shape[0] for p in pts]
    cat = torch.cat(pts)
    rr = torch.cat([p.new_tensor(OL.REGRESS_RANGES[i])[None].expand_as(p) for i, p in enumerate(pts)])
    for seed, n in ((0, 7), (1, 1), (2, 12)):
        gb, gl = _gt(seed, n)
        for cs in (True, False):
            a = OL.fcos_target_single(gb, gl, cat, rr, nums, STRIDES, cs, 1.5)
            lab_lvl, tgt_lvl, lab_img, tgt_img, gi = T.fcos_target(pts2, STRIDES, OL.REGRESS_RANGES, [gb], [gl], cs, 1.5)
            assert torch.equal(torch.cat(lab_lvl), a[0])
            assert torch.equal(torch.cat(tgt_lvl), a[1])
            assert torch.equal(gi[0], a[2])
            # brute force, point by point (sipmask_head.py:773-857 read literally)
            lvl_of = np.repeat(np.arange(5), nums)
            for p in range(0, cat.shape[0], 7):
                x, y = float(cat[p, 0]), float(cat[p, 1])
                best, lab = None, 0
                for k in range(n):
                    x1, y1, x2, y2 = [float(v) for v in gb[k]]
                    l, t, r, b = np.float32(x) - np.float32(x1), np.float32(y) - np.float32(y1), \
                        np.float32(x2) - np.float32(x), np.float32(y2) - np.float32(y)
                    if cs:
                        s = np.float32(STRIDES[lvl_of[p]] * 1.5)
                        cx, cy = (np.float32(x1) + np.float32(x2)) / 2, (np.float32(y1) + np.float32(y2)) / 2
                        ins = min(x - max(cx - s, x1), y - max(cy - s, y1), min(cx + s, x2) - x, min(cy + s, y2) - y) > 0
                    else:
                        ins = min(l, t, r, b) > 0
                    lo, hi = OL.REGRESS_RANGES[lvl_of[p]]
                    if ins and lo <= max(l, t, r, b) <= hi:
                        area = (np.float32(x2) - np.float32(x1) + 1) * (np.float32(y2) - np.float32(y1) + 1)
                        if best is None or area < best:
                            best, lab = area, int(gl[k])
                assert lab == int(a[0][p]), (seed, cs, p)
    # no ground truth: everything background (the reference itself breaks here, :779-781)
    lab, tgt, gi = T.assign_image(cat, cat[:, 0] * 0 + 8, cat[:, 0] * 0 - 1, cat[:, 0] * 0 + 64, torch.zeros(0, 4),
                                  torch.zeros(0, dtype=torch.long), True, 1.5)
    assert int(lab.sum()) == 0 and tgt.shape == (cat.shape[0], 4) and gi.numel() == 0


def test_oracle_deform_conv_backward_is_autograd_of_forward():
    torch.manual_seed(0)
    for (B, C, H, W, Co, G, s, p) in [(2, 8, 7, 9, 6, 4, 1, 1), (1, 4, 6, 5, 3, 1, 2, 1), (1, 8, 5, 5, 4, 2, 1, 0)]:
        x = torch.randn(B, C, H, W, dtype=torch.float64, requires_grad=True)
        w = torch.randn(Co, C, 3, 3, dtype=torch.float64, requires_grad=True)
        Ho, Wo = (H + 2 * p - 3) // s + 1, (W + 2 * p - 3) // s + 1
        off = (torch.randn(B, G * 18, Ho, Wo, dtype=torch.float64) * 1.5).requires_grad_()
        y = O.deform_conv(x, off, w, s, p, 1, G)
        go = torch.randn_like(y)
        gx, goff, gw = torch.autograd.grad(y, (x, off, w), go)
        rx, roff, rw = O.deform_conv_backward(x.detach(), off.detach(), w.detach(), go, s, p, 1, G)
        for a, b in ((gx, rx), (goff, roff), (gw, rw)):
            assert float((a - b).abs().max()) < 1e-12


def test_oracle_mask_loss_equals_numpy_crop_ops():
    g = torch.Generator().manual_seed(3)
    hm, wm, n, G = 20, 28, 9, 3
    fm = torch.randn(32, hm, wm, generator=g)
    cof = torch.randn(n, 128, generator=g) * 0.4
    xy = torch.rand(n, 2, generator=g) * torch.tensor([wm * 0.7, hm * 0.7])
    boxes = torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 12 + 1.5], 1)
    gtm = (torch.rand(G, hm, wm, generator=g) < 0.5).float()
    idx = torch.randint(0, G, (n,), generator=g)
    wgt = torch.rand(n, generator=g)
    loss, pre = OL.mask_loss_single(fm, cof, boxes, gtm, idx, wgt)
    img = fm.permute(1, 2, 0)
    probs = torch.stack([torch.sigmoid(img @ cof[:, 32 * q:32 * (q + 1)].t()) for q in range(4)], 0).numpy()
    pred = torch.from_numpy(O.crop_split(probs, boxes.numpy(), 2))
    gt = torch.from_numpy(O.crop_split_gt(gtm[idx].permute(1, 2, 0).contiguous().numpy(), boxes.numpy()))
    ref = F.binary_cross_entropy(pred, gt, reduction="none").sum(dim=(0, 1))
    ref = ref / (boxes[:, 2] - boxes[:, 0]) / (boxes[:, 3] - boxes[:, 1]) / n
    torch.testing.assert_close(pre, ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss, (ref * wgt).sum(), rtol=1e-5, atol=1e-6)


def test_benchmark_checkpoint_name_conversion():
    """B/ (maskrcnn-benchmark) parameter names -> mmdet names (pure host logic)."""
    from oracle import fcos_core as OB
    from oracle import model as OM
    from sipmask_amd.benchmark_variant import convert_state_dict
    head = OB.init_head_state_dict(seed=1)
    conv = convert_state_dict({"module." + k: v for k, v in head.items()})
    assert len(conv) == len(head)
    assert OM.tower_depths(conv) == (3, 4, True)
    assert torch.equal(conv["bbox_head.cls_convs.2.gn.weight"], head["rpn.head.cls_tower.7.weight"])
    assert torch.equal(conv["bbox_head.reg_convs.3.conv.bias"], head["rpn.head.bbox_tower.9.bias"])
    assert torch.equal(conv["bbox_head.fcos_cls.weight"], head["rpn.head.cls_logits.weight"])
    assert conv["bbox_head.scales.4.scale"].shape == ()
    trunk = {"backbone.body.stem.conv1.weight": torch.zeros(1), "backbone.body.layer3.5.bn2.running_var": torch.zeros(1),
             "backbone.body.layer2.0.downsample.0.weight": torch.zeros(1), "backbone.fpn.fpn_inner2.weight": torch.zeros(1),
             "backbone.fpn.fpn_layer4.bias": torch.zeros(1), "backbone.fpn.top_blocks.p6.weight": torch.zeros(1),
             "backbone.fpn.top_blocks.p7.bias": torch.zeros(1), "rpn.anchor_generator.cell_anchors.0": torch.zeros(1)}
    assert set(convert_state_dict(trunk)) == {
        "backbone.conv1.weight", "backbone.layer3.5.bn2.running_var", "backbone.layer2.0.downsample.0.weight",
        "neck.lateral_convs.0.conv.weight", "neck.fpn_convs.2.conv.bias", "neck.fpn_convs.3.conv.weight",
        "neck.fpn_convs.4.conv.bias"}


def test_oracle_ml_nms_equals_per_label_nms():
    """B/ ml_nms == the golden-pinned greedy NMS run per label (different labels never suppress each other)."""
    from oracle import fcos_core as OB
    rng = np.random.RandomState(2)
    n = 300
    xy = rng.rand(n, 2).astype(np.float32) * 200
    boxes = np.concatenate([xy, xy + rng.rand(n, 2).astype(np.float32) * 60 + 5], 1)
    scores = rng.rand(n).astype(np.float32)
    labels = rng.randint(1, 6, n)
    keep = OB.ml_nms(boxes, scores, labels, 0.6)
    ref = []
    for c in range(1, 6):
        idx = np.nonzero(labels == c)[0]
        k = O.nms(np.concatenate([boxes[idx], scores[idx, None]], 1), 0.6, mode="gpu")
        ref.extend(idx[k].tolist())
    assert sorted(ref) == keep.tolist()


def test_oracle_input_pipeline_properties():
    """numpy restatement of Resize/Normalize/Pad: size rule, identity resize, constant images, zero padding."""
    from oracle import pipeline as OP
    from sipmask_amd.input_pipeline import rescale_size
    for (h, w) in ((480, 640), (375, 500), (900, 1400), (800, 1333), (1333, 800)):
        assert rescale_size(h, w, (1333, 800)) == OP.rescale_size(h, w, (1333, 800))
        nh, nw, f = OP.rescale_size(h, w, (1333, 800))
        assert max(nh, nw) <= 1333 and min(nh, nw) <= 800 and (max(nh, nw) == 1333 or min(nh, nw) == 800)
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    np.testing.assert_array_equal(OP.resize_bilinear_u8(img, 37, 53), img)              # same size: identity
    const = np.full((20, 30, 3), 77, np.uint8)
    assert (OP.resize_bilinear_u8(const, 47, 61) == 77).all()                            # constants stay constant
    out, meta = OP.prepare(const, (100, 60), mean=(70, 70, 70), std=(1, 1, 1))
    nh, nw = meta["img_shape"][:2]
    assert out.shape[1] % 32 == 0 and out.shape[2] % 32 == 0
    assert (out[:, :nh, :nw] == 7).all() and (out[:, nh:, :] == 0).all() and (out[:, :, nw:] == 0).all()
