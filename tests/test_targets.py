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
