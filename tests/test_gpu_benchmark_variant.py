"""GPU tests of the maskrcnn-benchmark front-end (B/ = SipMask-benchmark/): head variant, pair selection +
same-label NMS + mask assembly against oracle/fcos_core.py, and the checkpoint name conversion."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fcos_core as OB  # noqa: E402
from oracle import model as OM  # noqa: E402

BM = dict(pre_nms_thresh=0.05, pre_nms_top_n=1000, nms_thresh=0.6, post_top_n=100)


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _to_b_names(sd):
    """inverse of benchmark_variant.convert_state_dict for backbone + FPN (test helper)"""
    out = {}
    for k, v in sd.items():
        if k.startswith("backbone.conv1") or k.startswith("backbone.bn1"):
            out["backbone.body.stem." + k[len("backbone."):]] = v
        elif k.startswith("backbone.layer"):
            out["backbone.body." + k[len("backbone."):]] = v
        elif k.startswith("neck.lateral_convs."):
            i, rest = k[len("neck.lateral_convs."):].split(".conv.")
            out["backbone.fpn.fpn_inner%d.%s" % (int(i) + 2, rest)] = v
        elif k.startswith("neck.fpn_convs."):
            i, rest = k[len("neck.fpn_convs."):].split(".conv.")
            i = int(i)
            out[("backbone.fpn.fpn_layer%d.%s" % (i + 2, rest)) if i < 3 else ("backbone.fpn.top_blocks.p%d.%s" % (i + 3, rest))] = v
    return out


def test_benchmark_head_and_postprocess_vs_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sipmask_amd.benchmark_variant import convert_state_dict
    from sipmask_amd.engine import SipMaskEngine
    sd_b = OB.init_head_state_dict(seed=4)
    sd = convert_state_dict(sd_b)
    assert OM.tower_depths(sd) == (3, 4, True) and "bbox_head.feat_align.conv_adaption.bias" in sd
    g = torch.Generator().manual_seed(2)
    B = 2
    sizes = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]
    feats = [torch.randn(B, 256, h, w, generator=g).to(torch.bfloat16).float() for h, w in sizes]
    image_size, ori_wh = (190, 250), (300, 228)               # padded input 192x256
    sf = min(image_size[0] / ori_wh[1], image_size[1] / ori_wh[0])
    eng = SipMaskEngine(sd, B, (192, 256), 50, None, 81, "cuda", (8, 16, 32, 64, 128), (image_size[0], image_size[1], 3),
                        head_sizes=list(sizes), scale_factor=sf, benchmark=BM)
    eng.load_pyramid([f.cuda() for f in feats])
    eng.run_head(with_post=True)
    torch.cuda.synchronize()
    # ---- head parity (bf16 engine vs f32 oracle)
    ref = OB.head_forward(sd_b, feats)
    out = eng.head_outputs()
    for name, got_l, ref_l, b0 in zip(("cls", "bbox", "ctr", "cof"), out[:4], ref[:4], (-4.0, 0.0, 0.0, 0.0)):
        for l in range(5):
            assert got_l[l].shape == ref_l[l].shape
            assert _rel(got_l[l] - b0, ref_l[l] - b0) < 0.1, (name, l, _rel(got_l[l] - b0, ref_l[l] - b0))
    assert float(out[1][0].min()) >= 0.0                       # relu(scale(bbox_pred))
    assert _rel(out[4], ref[4]) < 0.05
    # ---- post-processing on the ENGINE's own head outputs: identical f32 inputs on both sides
    cls, bb, ctr, cof, fm = [[t.cpu().float() for t in x] if isinstance(x, list) else x.cpu().float() for x in out]
    res = eng.results()
    tot = 0
    for b in range(B):
        r = OB.postprocess_single([c[b] for c in cls], [x[b] for x in bb], [c[b] for c in ctr], [c[b] for c in cof],
                                  fm[b], image_size, ori_wh, **BM)
        n = int(res["ndet"][b])
        assert n == r["bbox"].shape[0], (n, r["bbox"].shape[0])
        tot += n
        det = res["det_bboxes"][b, :n].cpu().numpy()
        lab = res["det_labels"][b, :n].cpu().numpy() + 1
        # the reference order is unspecified (unsorted topk, boolean masks): compare as sets in score order
        og = np.lexsort((det[:, 0], -det[:, 4]))
        orf = np.lexsort((r["bbox"][:, 0].numpy(), -r["scores"].numpy()))
        np.testing.assert_allclose(det[og, 4], r["scores"].numpy()[orf], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(det[og, :4], r["bbox"].numpy()[orf], rtol=1e-6, atol=1e-5)
        np.testing.assert_array_equal(lab[og], r["labels"].numpy()[orf])
        if n:
            gm = res["masks"][b, :n].cpu().numpy()[og]
            up = r["up"][orf]
            rm = (up > 0.4).numpy()
            assert gm.shape[1:] == rm.shape[1:]
            diff = gm != rm
            assert int(diff.sum()) <= 5 and bool(((up - 0.4).abs().numpy()[diff] < 1e-4).all())
    assert tot > 20


def test_benchmark_checkpoint_conversion_end_to_end():
    """A B/-named checkpoint (backbone.body.*, backbone.fpn.*, rpn.head.*) through SipMaskBenchmark gives the same
    detections as the engine built from the equivalent mmdet-named weights; results carry the BoxList fields."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sipmask_amd.benchmark_variant import SipMaskBenchmark, convert_state_dict
    from sipmask_amd.engine import SipMaskEngine
    trunk = {k: v for k, v in OM.init_state_dict(50, seed=6).items() if not k.startswith("bbox_head.")}
    head_b = OB.init_head_state_dict(seed=6)
    ckpt = dict(_to_b_names(trunk))
    ckpt.update(head_b)
    conv = convert_state_dict({"module." + k: v for k, v in ckpt.items()})
    assert set(k for k in trunk if "num_batches_tracked" not in k) <= set(conv)
    model = SipMaskBenchmark(ckpt, depth=50, inference_th=0.05)
    img = torch.randn(1, 3, 192, 256, generator=torch.Generator().manual_seed(1)).cuda()
    res = model(img, image_sizes=[(190, 250)], img_metas=[(300, 228)])[0]
    n = res["bbox"].shape[0]
    assert 0 < n <= 100 and res["mask"].shape == (n, 1, 228, 300) and res["mask"].dtype == torch.uint8
    assert int(res["labels"].min()) >= 1 and int(res["labels"].max()) <= 80
    sd = dict(trunk)
    sd.update(convert_state_dict(head_b))
    sf = min(190 / 228, 250 / 300)
    eng = SipMaskEngine(sd, 1, (192, 256), 50, None, 81, "cuda", (8, 16, 32, 64, 128), (190, 250, 3), scale_factor=sf,
                        benchmark=BM)
    r = eng.run(img)
    # same weights through both naming schemes = the same launch plan on the same bits: GroupNorm statistics are
    # fixed-point integer sums (order independent), so the head outputs agree bit for bit and so do the detections
    eng_b = list(model._engines.values())[0]
    assert torch.equal(eng_b.cls_cof, eng.cls_cof) and torch.equal(eng_b.reg_out, eng.reg_out)
    assert torch.equal(eng_b.basis, eng.basis)
    assert int(r["ndet"][0]) == n
    a = r["det_bboxes"][0, :int(r["ndet"][0])].cpu()
    matched = 0
    for i in range(n):
        d = (a[:, :4] - res["bbox"][i].cpu()).abs().max(1)[0]
        matched += int(d.min() < 0.5)
    assert matched >= 0.9 * n, (matched, n)


def _synthetic_targets(g, num_imgs, img_h, img_w, n_gt):
    rng = np.random.RandomState(int(torch.randint(0, 10000, (1,), generator=g)))
    out = []
    yy, xx = np.mgrid[:img_h, :img_w]
    for _ in range(num_imgs):
        xy = rng.rand(n_gt, 2) * np.array([img_w * 0.6, img_h * 0.6])
        wh = rng.rand(n_gt, 2) * np.array([img_w * 0.5, img_h * 0.5]) + 10
        b = np.concatenate([xy, np.minimum(xy + wh, [img_w - 1, img_h - 1])], 1).astype(np.float32)
        m = np.zeros((n_gt, img_h, img_w), np.uint8)
        for k in range(n_gt):
            cx, cy, rx, ry = (b[k, 0] + b[k, 2]) / 2, (b[k, 1] + b[k, 3]) / 2, (b[k, 2] - b[k, 0]) / 2, (b[k, 3] - b[k, 1]) / 2
            m[k] = (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0
        out.append(dict(bbox=torch.from_numpy(b), labels=torch.from_numpy(rng.randint(1, 81, n_gt).astype(np.int64)), masks=m))
    return out


def test_benchmark_loss_vs_oracle():
    """SipMaskLossComputation (product side, B/...sipmask/loss.py:330-487) on caller-provided head outputs: the four
    losses and their gradients w.r.t. every head output against the CPU oracle's autograd (oracle.fcos_core.loss, itself
    pinned to the reference's own __call__ by fixture L_b_loss).  f32 both sides: 2e-4."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sipmask_amd.benchmark_train import SipMaskLossComputation, compute_locations
    g = torch.Generator().manual_seed(41)
    B, C = 2, 80
    sizes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    strides = (8, 16, 32, 64, 128)
    mk = lambda c, sc, sh: [(torch.randn(B, c, h, w, generator=g) * sc + sh) for h, w in sizes]
    cls, ctr, cof = mk(C, 1.5, -3.0), mk(1, 1.0, 0.0), mk(128, 0.3, 0.0)
    reg = [torch.rand(B, 4, h, w, generator=g) * 3 + 0.5 for h, w in sizes]                # stride units, > 0 (relu'd)
    fm = torch.randn(B, 32, 64, 80, generator=g)
    tg = _synthetic_targets(g, B, 128, 160, 5)
    leaves_r = [[t.clone().requires_grad_() for t in ts] for ts in (cls, reg, ctr, cof)] + [fm.clone().requires_grad_()]
    ref, labels = OB.loss(leaves_r[0], leaves_r[1], leaves_r[2], leaves_r[3], leaves_r[4], [t["bbox"] for t in tg],
                          [t["labels"] for t in tg], [t["masks"] for t in tg])
    assert int((labels > 0).sum()) > 20
    sum(ref.values()).backward()
    leaves_d = [[t.cuda().requires_grad_() for t in ts] for ts in (cls, reg, ctr, cof)] + [fm.cuda().requires_grad_()]
    ev = SipMaskLossComputation()
    loc = compute_locations(leaves_d[0], strides)
    out = dict(zip(("loss_cls", "loss_reg", "loss_centerness", "loss_mask"),
                   ev(loc, leaves_d[0], leaves_d[1], leaves_d[2], leaves_d[3], leaves_d[4], tg)))
    for k in out:
        a, b = float(out[k].detach()), float(ref[k].detach())
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (k, a, b)
    sum(out.values()).backward()
    flat = lambda L: [t for ts in L[:4] for t in ts] + [L[4]]
    for a, b in zip(flat(leaves_d), flat(leaves_r)):
        assert b.grad is not None and a.grad is not None
        scale = float(b.grad.abs().max()) + 1e-12
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 2e-4 * scale + 1e-7, (tuple(a.shape), scale)


def test_benchmark_head_training_forward_and_step_vs_oracle():
    """SipMaskBenchmarkHead in training mode (sipmask.py:142-190 on the row-tensor HIP autograd ops): loads the B/
    parameter names, outputs against the f32 oracle head (bbox_reg in stride units = oracle's eval output / stride),
    then loss + backward: every parameter receives a gradient close to the oracle's autograd (wiring bound as in
    test_head_training_step_vs_oracle: cosine > 0.98, relative error < 0.2)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sipmask_amd.benchmark_train import SipMaskBenchmarkModule
    sd_b = OB.init_head_state_dict(seed=6)
    with torch.no_grad():
        sd_b["rpn.head.cls_logits.bias"].fill_(-3.0)
    mod = SipMaskBenchmarkModule()
    missing = mod.head.load_state_dict({k[len("rpn.head."):]: v.clone() for k, v in sd_b.items()}, strict=True)
    mod = mod.cuda().train()
    g = torch.Generator().manual_seed(8)
    B = 2
    sizes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    strides = (8, 16, 32, 64, 128)
    feats = [torch.randn(B, 256, h, w, generator=g).to(torch.bfloat16).float() for h, w in sizes]
    tg = _synthetic_targets(g, B, 128, 160, 4)
    osd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd_b.items()}
    oref = OB.head_forward(osd, feats)
    oreg = [r / s for r, s in zip(oref[1], strides)]
    oloss, labels = OB.loss(oref[0], oreg, oref[2], oref[3], oref[4], [t["bbox"] for t in tg], [t["labels"] for t in tg],
                            [t["masks"] for t in tg])
    assert int((labels > 0).sum()) > 10
    sum(oloss.values()).backward()
    out = mod.head([f.cuda() for f in feats])
    for name, got_l, ref_l in zip(("cls", "bbox", "ctr", "cof"), out[:4], (oref[0], oreg, oref[2], oref[3])):
        for l, (a, b) in enumerate(zip(got_l, ref_l)):
            assert _rel(a.detach(), b.detach()) < 0.06, (name, l, _rel(a.detach(), b.detach()))
    assert _rel(out[4].detach(), oref[4].detach()) < 0.05
    loss = mod([f.cuda() for f in feats], tg)
    for k in loss:
        a, b = float(loss[k].detach()), float(oloss[k].detach())
        assert abs(a - b) <= 2e-2 * max(1.0, abs(b)), (k, a, b)
    sum(loss.values()).backward()
    bad = []
    for name, p in mod.head.named_parameters():
        ref = osd["rpn.head." + name].grad
        assert p.grad is not None, name
        if ref is None or float(ref.norm()) == 0.0:
            continue
        got = p.grad.cpu().float().reshape(ref.shape)
        err = float((got - ref).norm() / ref.norm())
        cos = float((got * ref).sum() / (got.norm() * ref.norm() + 1e-30))
        if err > 0.2 or cos < 0.98:
            bad.append((name, round(err, 3), round(cos, 4)))
    assert not bad, sorted(bad, key=lambda t: -t[1])[:40]
