"""GPU parity of the exact-f32 plan (csrc/conv_f32.hip + the f32 layout / pool / GroupNorm kernels): every kernel
against torch-CPU fp32 / the oracle on ARBITRARY f32 inputs (no bf16 pre-rounding), tolerance = f32 accumulation
order only (rtol 1e-5 of the tensor scale), and the whole plan against the oracle end to end:
north_star's "mask logits within 1e-3 of reference" is asserted here at a small shape and in
tests/test_gpu_baseline_shape.py at the BASELINE shape."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import model as OM  # noqa: E402
from oracle import ops as O  # noqa: E402


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def _rows(x, cpad=None):
    B, C, H, W = x.shape
    r = x.permute(0, 2, 3, 1).reshape(-1, C)
    if cpad and cpad != C:
        r = torch.cat([r, torch.zeros(r.shape[0], cpad - C)], 1)
    return r.contiguous()


def _close(got, ref, tol=2e-5):
    scale = float(ref.abs().max()) + 1e-30
    err = float((got - ref).abs().max())
    assert err <= tol * scale, (err, scale, err / scale)


def _run_conv_f32(x, w, bias, stride, pad, flags=0, residual=None, offset=None, G=0):
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    B, C, Hh, Ww = x.shape
    Co, _, k, _ = w.shape
    cpad = (C + 3) // 4 * 4
    xh = _rows(x, cpad).to(dev)
    wq, co_pad = H.prep_conv_weight_f32(w.to(dev), cpad)
    Ho, Wo = (Hh + 2 * pad - k) // stride + 1, (Ww + 2 * pad - k) // stride + 1
    res = None
    if residual is not None:
        flags |= _lib.SM_CONV_RES_ADD
        res = _rows(residual).to(dev)
    y = torch.full((B * Ho * Wo, Co), float("nan"), dtype=torch.float32, device=dev)
    d = H.make_conv_desc(B, [(Hh, Ww)], [(Ho, Wo)], [0], [0], cpad, Co, co_pad, k, stride, pad, cpad, Co,
                         flags=flags, res_cstride=Co, deform_groups=G)
    offr = None if offset is None else _rows(offset).to(dev)
    H.conv2d_f32(d, xh, offr, wq, None if bias is None else bias.to(dev), res, y)
    torch.cuda.synchronize()
    return y.view(B, Ho, Wo, Co).permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("cfg", [
    (2, 64, 17, 23, 128, 3, 1, 1),     # 128-cout tile, ragged M
    (1, 256, 13, 21, 256, 3, 1, 1),    # tower shape
    (2, 128, 20, 28, 64, 1, 1, 0),     # 64-cout tile, 1x1
    (2, 64, 21, 19, 5, 3, 1, 1),       # 32-cout tile, cout tail (reg + centerness)
    (2, 3, 37, 45, 64, 7, 2, 3),       # stem: cin padded 3 -> 4, K = 196 padded to 208
    (1, 512, 16, 12, 1024, 1, 2, 0),   # strided 1x1 (downsample)
    (2, 256, 9, 11, 208, 3, 1, 1),     # cls + cof width
    (1, 768, 10, 14, 512, 1, 1, 0),
    (1, 256, 25, 42, 256, 3, 2, 1),    # P6: stride-2 3x3
])
def test_conv_f32_vs_torch(cfg):
    from sipmask_amd import _lib
    B, Ci, Hh, Ww, Co, k, s, p = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, Ci, Hh, Ww, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x, w, b, s, p)
    _close(_run_conv_f32(x, w, b, s, p), ref)
    r = torch.randn_like(ref)
    _close(_run_conv_f32(x, w, b, s, p, flags=_lib.SM_CONV_RELU, residual=r), F.relu(ref + r))
    _close(_run_conv_f32(x, w, None, s, p, flags=_lib.SM_CONV_IN_RELU), F.conv2d(F.relu(x), w, None, s, p))


def test_conv_f32_multilevel_scale_and_nearest_residual():
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    B, C = 2, 64
    sizes = [(12, 20), (6, 10), (3, 5)]
    lv = H.Levels(B, sizes)
    xs = [torch.randn(B, C, h, w, generator=g) for h, w in sizes]
    w = torch.randn(5, C, 3, 3, generator=g) / 24
    bias = torch.randn(5, generator=g)
    scales = [1.5, 0.5, 2.0]
    x = torch.cat([_rows(t) for t in xs]).to(dev)
    wq, co_pad = H.prep_conv_weight_f32(w.to(dev))
    y = torch.zeros(lv.rows, 8, dtype=torch.float32, device=dev)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, C, 5, co_pad, 3, 1, 1, C, 8, scale_nch=4, level_scale=scales)
    H.conv2d_f32(d, x, None, wq, bias.to(dev), None, y)
    torch.cuda.synchronize()
    for l, (h, wd) in enumerate(sizes):
        ref = F.conv2d(xs[l], w, bias, 1, 1)
        ref[:, :4] *= scales[l]
        got = y[lv.row0[l]:lv.row0[l] + B * h * wd].view(B, h, wd, 8).permute(0, 3, 1, 2)[:, :5].cpu()
        _close(got, ref)
    coarse = torch.randn(B, 128, 7, 9, generator=g)
    fine = torch.randn(B, C, 13, 18, generator=g)
    wl = torch.randn(128, C, 1, 1, generator=g) / 8
    ref = F.conv2d(fine, wl) + F.interpolate(coarse, size=(13, 18), mode="nearest")
    wq, co_pad = H.prep_conv_weight_f32(wl.to(dev))
    y = torch.zeros(B * 13 * 18, 128, dtype=torch.float32, device=dev)
    d = H.make_conv_desc(B, [(13, 18)], [(13, 18)], [0], [0], C, 128, co_pad, 1, 1, 0, C, 128,
                         flags=_lib.SM_CONV_RES_NEAREST, res_cstride=128, res_sizes=[(7, 9)], res_row0=[0])
    H.conv2d_f32(d, _rows(fine).to(dev), None, wq, None, _rows(coarse).to(dev), y)
    torch.cuda.synchronize()
    _close(y.view(B, 13, 18, 128).permute(0, 3, 1, 2).cpu(), ref)


@pytest.mark.parametrize("shape", [(2, 256, 13, 17, 256), (1, 64, 9, 11, 64)])
def test_deform_conv_f32_vs_oracle(shape):
    """f32 samples, f32 products: the deformable conv within 2e-5 of the oracle's restatement of
    deform_conv_cuda_kernel.cu:85-115,191-243 (bf16 plan: 2e-2)."""
    B, C, Hh, Ww, Co = shape
    g = torch.Generator().manual_seed(11)
    G = 4
    x = torch.randn(B, C, Hh, Ww, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    off = torch.randn(B, G * 18, Hh, Ww, generator=g) * 1.5
    off[0, :, 0, 0] = 0.0            # exact-integer sample
    off[0, 0::2, 1, 1] = -2.0        # h_im == -1 for the top taps -> must be excluded (> -1)
    ref = O.deform_conv(x, off, w, 1, 1, 1, G)
    _close(_run_conv_f32(x, w, None, 1, 1, offset=off, G=G), ref)
    _close(_run_conv_f32(x, w, None, 1, 1, offset=torch.zeros_like(off), G=G), F.conv2d(x, w, None, 1, 1))


def test_groupnorm_maxpool_layout_f32():
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    B, C = 2, 256
    sizes = [(20, 33), (10, 17), (5, 9), (3, 5), (2, 3)]
    lv = H.Levels(B, sizes)
    xs = [torch.randn(B, C, h, w, generator=g) * 3 + 20.0 for h, w in sizes]     # mean >> std: the cancellation case
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    x = torch.cat([_rows(t) for t in xs]).to(dev)
    stats = torch.zeros(B * 5 * 32 * 2, dtype=torch.float64, device=dev)
    H.groupnorm_f32(x, x, gamma.to(dev), beta.to(dev), stats, lv, C, 32, 1e-5, True)
    torch.cuda.synchronize()
    for l, (h, w) in enumerate(sizes):
        ref = F.relu(F.group_norm(xs[l].double(), 32, gamma.double(), beta.double(), 1e-5)).float()
        got = x[lv.row0[l]:lv.row0[l] + B * h * w].view(B, h, w, C).permute(0, 3, 1, 2).cpu()
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-5)
    img = torch.randn(2, 3, 21, 30, generator=g)
    xh = torch.empty(2 * 21 * 30, 4, dtype=torch.float32, device=dev)
    H.nchw_to_nhwc_f32(img.to(dev), xh, 4)
    got = xh.view(2, 21, 30, 4).permute(0, 3, 1, 2).cpu()
    assert torch.equal(got[:, :3], img) and float(got[:, 3].abs().max()) == 0.0
    xf = torch.randn(2, 64, 21, 30, generator=g)
    y = torch.empty(2 * 11 * 15, 64, dtype=torch.float32, device=dev)
    H.maxpool3x3s2_f32(_rows(xf).to(dev), y, 2, 21, 30, 64)
    assert torch.equal(y.view(2, 11, 15, 64).permute(0, 3, 1, 2).cpu(), F.max_pool2d(xf, 3, 2, 1))


@pytest.fixture(scope="module")
def small_case():
    _dev()
    from sipmask_amd.engine import SipMaskEngine
    B, Hh, Ww = 2, 192, 256
    sd = OM.init_state_dict(50, 0)
    img = torch.randn(B, 3, Hh, Ww, generator=torch.Generator().manual_seed(1))
    feats = OM.backbone_forward(sd, img)
    pyr = OM.fpn_forward(sd, feats)
    out = OM.head_forward(sd, pyr)
    allc = torch.cat([c[0].reshape(-1) for c in out[0]]) - sd["bbox_head.fcos_cls.bias"][0]
    OM.calibrate_cls_bias(sd, allc, target=400)
    out = OM.head_forward(sd, pyr)
    eng = SipMaskEngine(sd, B, (Hh, Ww), 50, precision="f32")
    res = eng.run(img.cuda())
    torch.cuda.synchronize()
    return dict(sd=sd, img=img, feats=feats, pyr=pyr, out=out, eng=eng, res=res, B=B, hw=(Hh, Ww))


def test_f32_plan_stage_parity(small_case):
    """every stage of the f32 plan within 1e-4 (relative to the stage's largest value) of the fp32 oracle, from the
    SAME IMAGE; the bf16 plan's bound for the same stages is 2-15 % (tests/test_gpu_engine.py)"""
    c = small_case
    eng, B = c["eng"], c["B"]
    for i, (buf, h, w, ch) in enumerate(eng.backbone_feats):
        _close(buf.view(B, h, w, ch).permute(0, 3, 1, 2).cpu(), c["feats"][i], 1e-4)
    lv = eng.lv
    for l, (h, w) in enumerate(lv.sizes):
        _close(eng.pyr[lv.row0[l]:lv.row0[l] + B * h * w].view(B, h, w, 256).permute(0, 3, 1, 2).cpu(), c["pyr"][l], 1e-4)
    cls, bb, ctr, cof, fm = eng.head_outputs()
    ocls, obb, octr, ocof, ofm = c["out"]
    for l in range(5):
        _close(cls[l].cpu(), ocls[l], 2e-4)
        _close(bb[l].cpu(), obb[l], 2e-4)
        _close(ctr[l].cpu(), octr[l], 2e-4)
        _close(cof[l].cpu(), ocof[l], 2e-4)
    _close(fm.cpu(), ofm, 2e-4)


def _engine_det_keys(eng, res, b):
    """(level, position, label) of the engine's detections of image b: keep = candidate slot; level l owns the slots
    [cand0_l, cand0_l + min(nms_pre, h_l*w_l)), cand_pos = position inside the level"""
    n = int(res["ndet"][b])
    keep = res["idxs_keep"][b, :n].cpu().long()
    pos = eng.sel["cand_pos"][b].cpu().long()[keep]
    bounds = np.cumsum([0] + [min(eng.cfg["nms_pre"], h * w) for h, w in eng.lv.sizes])
    lev = np.searchsorted(bounds, keep.numpy(), side="right") - 1
    lab = res["det_labels"][b, :n].cpu().numpy()
    return [(int(l), int(p), int(c)) for l, p, c in zip(lev, pos.numpy(), lab)]


def test_f32_plan_mask_logits_and_detections(small_case):
    """north_star: mask logits within 1e-3 of the reference on identical inputs.  Here from the same IMAGE through
    ResNet-50 + FPN + head (f32 plan), against the oracle's own post-processing.  Logits that differ by f32 rounding
    can swap two near-equal ranking keys, so detections are compared as sets of (level, position, label): at most 2
    per image may differ (the bit-exact keep-index tests on IDENTICAL inputs are in test_gpu_kernels.py); masks of the
    common detections may differ only where the upsampled probability is within 1e-3 of the threshold."""
    c = small_case
    eng, res, B = c["eng"], c["res"], c["B"]
    Hh, Ww = c["hw"]
    ocls, obb, octr, ocof, ofm = c["out"]
    total = 0
    for b in range(B):
        r = OM.get_masks_single([x[b] for x in ocls], [x[b] for x in obb], [x[b] for x in octr], [x[b] for x in ocof],
                                ofm[b], (Hh, Ww, 3), OM.DEFAULT_TEST_CFG)
        n = r["det_bboxes"].shape[0]
        total += n
        assert abs(int(res["ndet"][b]) - n) <= 2
        if not n:
            continue
        keep = torch.from_numpy(r["idxs_keep"])
        okeys = [(int(l), int(p), int(c)) for l, p, c in zip(r["cand_level"][keep], r["cand_pos"][keep], r["det_labels"])]
        gkeys = _engine_det_keys(eng, res, b)
        common = set(okeys) & set(gkeys)
        assert len(common) >= n - 2, (n, len(common))
        # mask logits of the ORACLE's kept detections: engine basis . engine coefficients vs the oracle's
        lv = eng.lv
        rows = torch.tensor([lv.row0[l] + b * lv.sizes[l][0] * lv.sizes[l][1] + q for l, q, _ in okeys])
        gcof = eng.cls_cof[rows.cuda()][:, eng.ncls:].cpu()
        gbasis = eng.basis.view(B, eng.hm * eng.wm, 32)[b].cpu()
        obasis = ofm[b].permute(1, 2, 0).reshape(-1, 32)
        worst = 0.0
        for q in range(4):
            got = gbasis @ gcof[:, 32 * q:32 * (q + 1)].t()
            ref = obasis @ r["det_cofs"][:, 32 * q:32 * (q + 1)].t()
            worst = max(worst, float((got - ref).abs().max()))
        assert worst <= 1e-3, worst
        gi = {k: i for i, k in enumerate(gkeys)}
        for i, k in enumerate(okeys):
            if k not in gi:
                continue
            j = gi[k]
            np.testing.assert_allclose(res["det_bboxes"][b, j].cpu().numpy(), r["det_bboxes"][i], rtol=1e-4, atol=2e-3)
            diff = res["masks"][b, j].cpu() != r["masks"][i]
            assert bool(((r["up"][i] - 0.4).abs()[diff] < 1e-3).all())
            assert int(diff.sum()) <= max(3, int(1e-4 * diff.numel()))
    assert total > 0
