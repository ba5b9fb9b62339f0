"""The split-precision ("x3") head plan: binary16 halves [hi | lo | hi] of f32 activations against weights
[hi | hi | lo] on the bf16 plan's MFMA kernels with v_mfma_f32_32x32x16_f16 (SM_CONV_F16), f32 activations between the
layers.  Kernel tests against torch float64 convolutions of the SAME f32 inputs (no operand pre-rounding: the point of
the plan is that f32 operands survive), plan tests against the fp32 oracle.  Reference arithmetic: the reference head
is fp32 end to end, M/mmdet/models/anchor_heads/sipmask_head.py:241-287,609-633."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import model as OM  # noqa: E402


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def _rows(ts):
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts]).contiguous()


def test_split3_layout_and_precision():
    """sm_split3_f16: hi = f16(v) exactly, hi + lo reproduces v to 2^-21 relative (|v| in binary16's normal range), from
    f32 and from bf16 rows, into a channel slice of a wider destination; out-of-range values saturate"""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(1000, 64, generator=g) * torch.logspace(-3, 2, 64)).contiguous()
    x[0, :4] = torch.tensor([0.0, -0.0, 7e4, -1e9])
    y = torch.full((1000, 3 * 96), 7.0, dtype=torch.float16, device=dev)
    H.split3_f16(x.to(dev), y, 64, 96, 16)
    yc = y.cpu().float()
    hi, lo, hi2 = yc[:, 16:80], yc[:, 96 + 16:96 + 80], yc[:, 192 + 16:192 + 80]
    xc = x.clamp(-65504, 65504)
    assert torch.equal(hi, xc.half().float()) and torch.equal(hi, hi2)
    # |lo| <= 2^-11 |v|; it is a normal binary16 (11 more bits: 2^-22 |v| in all) while |lo| >= 2^-14, i.e. |v| >= 0.125,
    # and is quantised to binary16's subnormal step 2^-24 below that (absolute error <= 2^-25)
    big = xc.abs() >= 0.25
    assert float(((hi + lo - xc).abs() / xc.abs().clamp_min(1e-30))[big].max()) <= 2.0 ** -21
    assert float((hi + lo - xc).abs()[~big].max()) <= 2.0 ** -25
    assert bool((yc[:, :16] == 7).all()) and bool((yc[:, 80:96] == 7).all())   # other sources' slices untouched
    xb = x.to(torch.bfloat16)
    y2 = torch.empty(1000, 192, dtype=torch.float16, device=dev)
    H.split3_f16(xb.to(dev), y2)
    y2 = y2.cpu().float()
    xbc = xb.float().clamp(-65504, 65504)
    assert torch.equal(y2[:, :64], xbc.half().float()) and torch.equal(y2[:, 128:], y2[:, :64])
    assert float((y2[:, :64] + y2[:, 64:128] - xbc).abs().max()) <= float(xbc.abs().max()) * 2.0 ** -21


def test_paired_layout_split_and_groupnorm():
    """Round 6, the PAIRED operand layout (sm_conv_desc.x3_pairs): per 16 channels [hi 16 | lo 16].  sm_split_pairs_f16 writes
    exactly the halves sm_split3_f16 writes, from f32 and from bf16 rows, also into a channel slice of a wider destination;
    sm_groupnorm_apply_x3p writes exactly the halves (and the in-place f32 rows) sm_groupnorm_apply_x3 writes."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(777, 64, generator=g) * torch.logspace(-3, 2, 64)).contiguous()
    for src in (x, x.to(torch.bfloat16)):
        y3 = torch.empty(777, 3 * 96, dtype=torch.float16, device=dev)
        yp = torch.full((777, 2 * 96), 7.0, dtype=torch.float16, device=dev)
        H.split3_f16(src.to(dev), y3, 64, 96, 16)
        H.split_pairs_f16(src.to(dev), yp, 64, 96, 16)
        hi, lo = H.pairs_to_float(yp.cpu(), 96)
        assert torch.equal(hi[:, 16:80], y3[:, 16:80].cpu().float()) and torch.equal(lo[:, 16:80], y3[:, 96 + 16:96 + 80].cpu().float())
        assert bool((hi[:, :16] == 7).all()) and bool((lo[:, 80:] == 7).all())       # other sources' slices untouched
    B, sizes, C = 2, [(9, 14), (5, 7)], 256
    lv = H.Levels(B, sizes)
    v = torch.randn(lv.rows, C, generator=g).to(dev) * 3 + 0.5
    gam, bet = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    stats = H.gn_stats_alloc(B * len(sizes) * 32, dev)
    H.gn_stats_f32_fix(v, stats, lv, C, 32)
    a32, b32 = v.clone(), v.clone()
    y3 = torch.empty(lv.rows, 3 * C, dtype=torch.float16, device=dev)
    yp = torch.empty(lv.rows, 2 * C, dtype=torch.float16, device=dev)
    H.groupnorm_apply_x3(a32, gam, bet, stats, lv, C, 32, 1e-5, True, y_f32=a32, y_split=y3)
    H.groupnorm_apply_x3(b32, gam, bet, stats, lv, C, 32, 1e-5, True, y_f32=b32, y_pairs=yp)
    torch.cuda.synchronize()
    hi, lo = H.pairs_to_float(yp.cpu(), C)
    assert torch.equal(a32, b32) and torch.equal(hi, y3[:, :C].cpu().float()) and torch.equal(lo, y3[:, C:2 * C].cpu().float())
    assert float(hi.abs().max()) > 0 and float(lo.abs().max()) > 0


@pytest.mark.parametrize("kernel", ["igemm", "igemm256", "patch", "patch_pairs"])
def test_x3_conv_matches_f64_conv_of_f32_operands(kernel):
    """3x3 256->256 over a 3-level pyramid, grouped (2 weight sets, shared input), fused fixed-point GN statistics, f32
    output: within 4e-6 of the float64 convolution of the SAME f32 operands, relative to the output's largest value
    (bf16 operands: 4e-3; K = 2304 products of ~2^-21 each plus the f32 accumulation).  The weight scale is a power of two and is undone exactly."""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    B, C, G = 2, 256, 2
    pairs = kernel == "patch_pairs"       # round 6: paired operands, three products on fragments read once (x3_pairs)
    sizes = [(40, 66), (20, 33), (5, 9)] if not kernel.startswith("patch") else [(100, 168), (50, 84), (25, 42)]
    lv = H.Levels(B, sizes)
    xs = [torch.randn(B, C, h, w, generator=g) * 1.7 for h, w in sizes]
    ws = [torch.randn(C, C, 3, 3, generator=g) * 0.03 for _ in range(G)]
    bias = [torch.randn(C, generator=g) for _ in range(G)]
    nk = 2 if pairs else 3
    x3 = torch.empty(lv.rows, nk * C, dtype=torch.float16, device=dev)
    (H.split_pairs_f16 if pairs else H.split3_f16)(_rows(xs).to(dev), x3)
    scale = H.x3_weight_scale(ws)
    assert scale == 2.0 ** round(np.log2(scale)) and 2048 <= max(float(w.abs().max()) for w in ws) * scale < 4096
    flags = _lib.SM_CONV_F16 | _lib.SM_CONV_OUT_F32
    if pairs:
        packed = [H.prep_conv_weight_patch_x3p(w.to(dev), scale)[0] for w in ws]
        co_pad = 256
    elif kernel == "patch":
        packed = [H.prep_conv_weight_patch_x3(w.to(dev), scale)[0] for w in ws]
        co_pad = 256
    else:
        packed = [H.prep_conv_weight_x3(w.to(dev), scale) for w in ws]
        co_pad = packed[0][1]
        packed = [p[0] for p in packed]
        if kernel == "igemm256":
            flags |= _lib.SM_CONV_DBG_TILE256 | _lib.SM_CONV_DBG_HAND_PLACED | _lib.SM_CONV_DBG_BIG_TILES
    assert packed[0].dtype == torch.float16
    wq = torch.stack(packed).contiguous()
    S = 2 * B * len(sizes) * (C // 8)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, nk * C, C, co_pad, 3, 1, 1, nk * C, C, flags=flags, ngroups=G,
                         x_group_rows=0, y_group_rows=lv.rows, w_group_stride=packed[0].numel(), bias_group_stride=C,
                         gn_group_stride=S, acc_scale=1.0 / scale, x3_pairs=int(pairs))
    y = torch.zeros(G * lv.rows, C, dtype=torch.float32, device=dev)
    stats = torch.full((G * S,), 5, dtype=torch.int64, device=dev)
    bq = torch.stack(bias).to(dev).contiguous()
    if kernel.startswith("patch"):
        assert H.conv3x3_patch_supported(d)
        H.conv3x3_patch(d, x3, wq, bq, y, stats)
    else:
        H.conv2d_gn_stats(d, x3, None, wq, bq, None, y, stats)
    torch.cuda.synchronize()
    for gi in range(G):
        st = H.gn_stats_to_float(stats[gi * S:(gi + 1) * S].view(B, len(sizes), C // 8, 2).cpu())
        for l, (h, w) in enumerate(sizes):
            ref = F.conv2d(xs[l].double(), ws[gi].double(), bias[gi].double(), 1, 1)
            r0 = gi * lv.rows + lv.row0[l]
            got = y[r0:r0 + B * h * w].view(B, h, w, C).permute(0, 3, 1, 2).cpu().double()
            err = float((got - ref).abs().max()) / float(ref.abs().max())
            assert err < 4e-6, (kernel, gi, l, err)
            r8 = ref.reshape(B, C // 8, 8 * h * w)
            torch.testing.assert_close(st[:, l, :, 0], r8.sum(-1), rtol=1e-5, atol=1e-3 * (h * w) ** 0.5)
            torch.testing.assert_close(st[:, l, :, 1], (r8 * r8).sum(-1), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("kernel", ["igemm", "patch"])
def test_two_term_conv_on_bf16_rows_equals_the_three_term_conv(kernel):
    """Round 5: a bf16 source has no low half, so [hi | hi] x [w_hi | w_lo] (sm_split2_f16 + prep(..., terms=2): 2/3 of the MFMA
    work) must give what [hi | lo | hi] x [w_hi | w_hi | w_lo] gives -- the same f32 accumulation without the products that are
    zero.  Grouped launch with shared input and fused GroupNorm statistics, as the x3 plan's first tower launch runs it."""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    B, C, G = 2, 256, 2
    sizes = [(40, 66), (20, 33), (5, 9)] if kernel != "patch" else [(100, 168), (50, 84), (25, 42)]
    lv = H.Levels(B, sizes)
    xs = [(torch.randn(B, C, h, w, generator=g) * 1.7).to(torch.bfloat16) for h, w in sizes]
    ws = [torch.randn(C, C, 3, 3, generator=g) * 0.03 for _ in range(G)]
    bias = torch.stack([torch.randn(C, generator=g) for _ in range(G)]).to(dev).contiguous()
    rows = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, C) for x in xs]).contiguous().to(dev)
    scale = H.x3_weight_scale(ws)
    outs = {}
    for terms in (3, 2):
        xq = torch.empty(lv.rows, terms * C, dtype=torch.float16, device=dev)
        (H.split3_f16 if terms == 3 else H.split2_f16)(rows, xq)
        if terms == 2:
            # hi == the bf16 value wherever binary16 has it (normals: >= 2^-14; subnormals keep multiples of 2^-24)
            hi_f, x_f = xq[:, :C].float(), rows.float()
            assert torch.equal(xq[:, :C], xq[:, C:]) and float((hi_f - x_f).abs().max()) <= 2.0 ** -25
            assert torch.equal(hi_f[x_f.abs() >= 2.0 ** -14], x_f[x_f.abs() >= 2.0 ** -14])
        if kernel == "patch":
            packed = [H.prep_conv_weight_patch_x3(w.to(dev), scale, terms=terms)[0] for w in ws]
            co_pad = 256
        else:
            pk = [H.prep_conv_weight_x3(w.to(dev), scale, terms) for w in ws]
            co_pad, packed = pk[0][1], [p[0] for p in pk]
        wq = torch.stack(packed).contiguous()
        S = 2 * B * len(sizes) * (C // 8)
        d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, terms * C, C, co_pad, 3, 1, 1, terms * C, C,
                             flags=_lib.SM_CONV_F16 | _lib.SM_CONV_OUT_F32, ngroups=G, x_group_rows=0, y_group_rows=lv.rows,
                             w_group_stride=packed[0].numel(), bias_group_stride=C, gn_group_stride=S, acc_scale=1.0 / scale)
        y = torch.zeros(G * lv.rows, C, dtype=torch.float32, device=dev)
        stats = torch.zeros(G * S, dtype=torch.int64, device=dev)
        if kernel == "patch":
            assert H.conv3x3_patch_supported(d)
            H.conv3x3_patch(d, xq, wq, bias, y, stats)
        else:
            H.conv2d_gn_stats(d, xq, None, wq, bias, None, y, stats)
        torch.cuda.synchronize()
        outs[terms] = (y, stats)
    y3, y2 = outs[3][0], outs[2][0]
    # (a bf16 value below 2^-17 can leave a low half of one binary16 ulp, 2^-24: nothing a randn draw of this size contains)
    assert float((y3 - y2).abs().max()) <= 1e-6 * float(y3.abs().max())
    assert float((y3 != y2).float().mean()) < 1e-3
    ref = F.conv2d(xs[0].double(), ws[1].double(), bias[1].cpu().double(), 1, 1)
    h, w = sizes[0]
    got = y2[lv.rows:lv.rows + B * h * w].view(B, h, w, C).permute(0, 3, 1, 2).cpu().double()
    assert float((got - ref).abs().max()) / float(ref.abs().max()) < 4e-6


def test_x3_plan_first_tower_launch_runs_two_terms_on_the_bf16_pyramid(head_case, monkeypatch):
    """The x3 plan's first tower launch reads the bf16 FPN outputs as [hi | hi] (mode "x2"); switched back to three terms
    the head outputs and detections do not move (<= 2e-6 of the tensor's largest value); a head-only plan fed f32 features
    keeps three terms."""
    import sipmask_amd.engine as E
    from sipmask_amd.engine import SipMaskEngine
    sd = head_case["sd"]
    img = torch.randn(2, 3, 192, 256, generator=torch.Generator().manual_seed(4)).cuda()
    res = {}
    for two in (True, False):
        monkeypatch.setattr(E, "_X3_TOWER0_TWO_TERMS", two)
        eng = SipMaskEngine(sd, 2, (192, 256), 50, precision="head_x3")
        t0 = [c for c in eng.convs if c.name == "head.tower0"][0]
        assert t0.mode == ("x2" if two else ("x3p" if E._X3_PAIRS else "x3")) and t0.mfma_flops == t0.flops * (2 if two else 3)
        assert all(c.mode != "x2" for c in eng.convs if c.name != "head.tower0")
        assert [c.mode for c in eng.convs if c.name == "head.tower1"] == ["x3p" if E._X3_PAIRS else "x3"]
        r = eng.run(img)
        torch.cuda.synchronize()
        res[two] = (eng.cls_cof.clone(), eng.reg_out.clone(), {k: v.clone() for k, v in r.items()})
    # (f32 accumulation order only: the two-term launch walks K as [hi x w_hi | hi x w_lo] per 32 channels, the paired
    # three-term launch per 16 channels with the zero lo products in between -- measured 1.04e-6 of the largest value)
    for a, b in zip(res[True][:2], res[False][:2]):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    for k in ("ndet", "det_labels", "idxs_keep"):
        assert torch.equal(res[True][2][k], res[False][2][k]), k
    sizes = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]
    hsd = {k: v for k, v in sd.items() if k.startswith("bbox_head.")}
    monkeypatch.setattr(E, "_X3_TOWER0_TWO_TERMS", True)
    heng = SipMaskEngine.for_head(hsd, 2, sizes, img_shape=(192, 256, 3), precision="head_x3")
    assert all(c.mode != "x2" for c in heng.convs)


def test_x3_small_cout_and_1x1_convs():
    """the head's other x3 launches: 1x1 over 3*768 channels (sip_mask_lat0), 3x3 to 32 couts (sip_mask_lat) and to 8
    couts with per-level Scale on 4 of them (fcos_reg + centerness), ReLU, bias -- f32 out, within 4e-6 of float64"""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    B = 2
    for (ci, co, k, sizes, relu, nch) in [(768, 512, 1, [(25, 40)], True, 0), (512, 32, 3, [(25, 40)], True, 0),
                                          (256, 5, 3, [(20, 33), (10, 17), (5, 9)], False, 4)]:
        lv = H.Levels(B, sizes)
        xs = [torch.randn(B, ci, h, w, generator=g).abs() for h, w in sizes]
        wt = torch.randn(co, ci, k, k, generator=g) * (0.5 / (ci * k * k) ** 0.5)
        bias = torch.randn(co, generator=g)
        lscale = [1.0 + 0.25 * l for l in range(len(sizes))]
        x3 = torch.empty(lv.rows, 3 * ci, dtype=torch.float16, device=dev)
        H.split3_f16(_rows(xs).to(dev), x3)
        scale = H.x3_weight_scale([wt])
        wq, co_pad = H.prep_conv_weight_x3(wt.to(dev), scale)
        cs = (co + 7) // 8 * 8
        d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 3 * ci, co, co_pad, k, 1, k // 2, 3 * ci, cs,
                             flags=_lib.SM_CONV_F16 | _lib.SM_CONV_OUT_F32 | (_lib.SM_CONV_RELU if relu else 0), scale_nch=nch,
                             level_scale=lscale, acc_scale=1.0 / scale)
        y = torch.zeros(lv.rows, cs, dtype=torch.float32, device=dev)
        H.conv2d(d, x3, wq, bias.to(dev), None, y)
        torch.cuda.synchronize()
        for l, (h, w) in enumerate(sizes):
            ref = F.conv2d(xs[l].double(), wt.double(), bias.double(), 1, k // 2)
            if nch:
                ref[:, :nch] *= lscale[l]
            if relu:
                ref = ref.clamp_min(0)
            got = y[lv.row0[l]:lv.row0[l] + B * h * w, :co].view(B, h, w, co).permute(0, 3, 1, 2).cpu().double()
            err = float((got - ref).abs().max()) / float(ref.abs().max())
            assert err < 4e-6, (ci, co, k, l, err)
    # binary16 operands need f32 output, no residual
    d.flags &= ~_lib.SM_CONV_OUT_F32
    with pytest.raises(RuntimeError):
        H.conv2d(d, x3, wq, None, None, y)


def test_paired_operands_on_the_1x1_and_small_cout_kernels():
    """Round 6: sm_conv_desc.x3_pairs beyond the patch kernel -- sm_conv2d's 32-wide-K kernel (sip_mask_lat0's 1x1 convs by
    linearity, 256 -> 512), sm_conv3x3_smallco (sip_mask_lat 512 -> 32; fcos_reg + centerness 256 -> 5 with per-level Scale
    over three levels) and sm_upsample_sum2 writing the paired layout: every conv within 4e-6 of float64 and within 2e-6 of the
    K-concatenated launch it replaces; the up-sum's halves bit-equal to the [hi | lo | hi] output's."""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    g = torch.Generator().manual_seed(17)
    B = 2
    F16F = _lib.SM_CONV_F16 | _lib.SM_CONV_OUT_F32
    for (ci, co, k, sizes, relu, nch) in [(256, 512, 1, [(25, 40)], False, 0), (512, 32, 3, [(25, 40)], True, 0),
                                          (256, 5, 3, [(20, 33), (10, 17), (5, 9)], False, 4)]:
        lv = H.Levels(B, sizes)
        xs = [torch.randn(B, ci, h, w, generator=g).abs() for h, w in sizes]
        wt = torch.randn(co, ci, k, k, generator=g) * (0.5 / (ci * k * k) ** 0.5)
        bias = torch.randn(co, generator=g)
        lscale = [1.0 + 0.25 * l for l in range(len(sizes))]
        scale = H.x3_weight_scale([wt])
        cs = (co + 7) // 8 * 8
        outs = {}
        for pairs in (False, True):
            nk = 2 if pairs else 3
            xq = torch.empty(lv.rows, nk * ci, dtype=torch.float16, device=dev)
            (H.split_pairs_f16 if pairs else H.split3_f16)(_rows(xs).to(dev), xq)
            y = torch.zeros(lv.rows, cs, dtype=torch.float32, device=dev)
            kw = dict(flags=F16F | (_lib.SM_CONV_RELU if relu else 0), scale_nch=nch, level_scale=lscale, acc_scale=1.0 / scale,
                      x3_pairs=int(pairs))
            if k == 1:
                wq, co_pad = (H.prep_conv_weight_x3p if pairs else H.prep_conv_weight_x3)(wt.to(dev), scale)
                d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, nk * ci, co, co_pad, 1, 1, 0, nk * ci, cs, **kw)
                assert H.conv_plan(d)["k_step"] == 32
                H.conv2d(d, xq, wq, bias.to(dev), None, y)
            else:
                wpad = torch.cat([wt, torch.zeros(cs - co, ci, 3, 3)]) if cs != co else wt
                bpad = torch.cat([bias, torch.zeros(cs - co)]) if cs != co else bias
                wq = H.prep_conv_weight_smallco(wpad.to(dev), x3_scale=scale, pairs=pairs)
                d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, nk * ci, cs, 32, 3, 1, 1, nk * ci, cs, **kw)
                assert H.conv3x3_smallco_supported(d)
                H.conv3x3_smallco(d, xq, wq, bpad.to(dev), y)
            torch.cuda.synchronize()
            outs[pairs] = y
        top = 0.0
        for l, (h, w) in enumerate(sizes):
            ref = F.conv2d(xs[l].double(), wt.double(), bias.double(), 1, k // 2)
            if nch:
                ref[:, :nch] *= lscale[l]
            if relu:
                ref = ref.clamp_min(0)
            got = outs[True][lv.row0[l]:lv.row0[l] + B * h * w, :co].view(B, h, w, co).permute(0, 3, 1, 2).cpu().double()
            top = max(top, float(ref.abs().max()))
            err = float((got - ref).abs().max()) / float(ref.abs().max())
            assert err < 4e-6, (ci, co, k, l, err)
        assert float((outs[True] - outs[False]).abs().max()) <= 2e-6 * top, (ci, co, k)
    # a bf16-plan descriptor (no SM_CONV_F16) must refuse the field
    d.flags = _lib.SM_CONV_OUT_F32
    assert not H.conv3x3_smallco_supported(d)
    # sm_upsample_sum2 with the paired output
    h0, w0, C = 24, 40, 512
    n0, n1, n2 = B * h0 * w0, B * (h0 // 2) * (w0 // 2), B * (h0 // 4) * (w0 // 4)
    a0, a1, a2 = (torch.randn(n, C, generator=g).to(dev) for n in (n0, n1, n2))
    o3 = torch.empty(n0, 3 * C, dtype=torch.float16, device=dev)
    op = torch.empty(n0, 2 * C, dtype=torch.float16, device=dev)
    H.upsample_sum2(a0, a1, a2, o3, B, h0, w0, C, relu=True)
    H.upsample_sum2(a0, a1, a2, op, B, h0, w0, C, relu=True)
    torch.cuda.synchronize()
    hi, lo = H.pairs_to_float(op.cpu(), C)
    assert torch.equal(hi, o3[:, :C].cpu().float()) and torch.equal(lo, o3[:, C:2 * C].cpu().float())


def test_fused_split_producers():
    """the two producers that write a split operand directly: sm_upsample_bilinear_x3 (the mask branch's [l0 | up2(l1) |
    up4(l2)] concatenation) and the SM_CONV_OUT_X3 epilogue (sip_mask_lat0 -> sip_mask_lat without an f32 round trip);
    the conv's split output equals "f32 output, then sm_split3_f16" bit for bit (same accumulators); the upsampling kernel
    agrees with the f32 upsampling kernel within an ulp (the compiler contracts the bilinear expression differently)"""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    g = torch.Generator().manual_seed(13)
    B, C = 2, 256
    sizes = [(24, 40), (12, 20), (6, 10)]
    lv = H.Levels(B, sizes)
    xs = [torch.randn(B, C, h, w, generator=g) for h, w in sizes]
    rows = _rows(xs).to(dev)
    n0 = B * 24 * 40
    cat3 = torch.zeros(n0, 3 * 768, dtype=torch.float16, device=dev)
    cat32 = torch.zeros(n0, 768, dtype=torch.float32, device=dev)
    for l, (h, w) in enumerate(sizes):
        src = rows[lv.row0[l]:lv.row0[l] + B * h * w]
        H.upsample_bilinear_x3(src, cat3, B, h, w, C, 2 ** l, 768, 256 * l)
        H.upsample_bilinear(src, cat32, B, h, w, C, 2 ** l, C, 768, 256 * l, True)
    ref3 = torch.empty_like(cat3)
    H.split3_f16(cat32, ref3, 768)
    torch.cuda.synchronize()
    rec3 = cat3[:, :768].float() + cat3[:, 768:1536].float()
    torch.testing.assert_close(rec3, cat32, rtol=3e-7, atol=1e-7)
    assert torch.equal(cat3[:, :768], cat3[:, 1536:])
    cat3 = ref3                                        # the conv below reads the reference split of the f32 concatenation
    up = torch.cat([xs[0]] + [F.interpolate(xs[l], scale_factor=2 ** l, mode="bilinear", align_corners=False) for l in (1, 2)], 1)
    rec = (cat3[:, :768].float() + cat3[:, 768:1536].float()).view(B, 24, 40, 768).permute(0, 3, 1, 2).cpu()
    torch.testing.assert_close(rec, up, rtol=1e-5, atol=1e-5)
    # 1x1 conv 768 -> 512 + bias + ReLU with the split output
    wt = torch.randn(512, 768, 1, 1, generator=g) * 0.05
    bias = torch.randn(512, generator=g)
    scale = H.x3_weight_scale([wt])
    wq, co_pad = H.prep_conv_weight_x3(wt.to(dev), scale)
    one = [(24, 40)]
    l1 = H.Levels(B, one)
    base = _lib.SM_CONV_F16 | _lib.SM_CONV_RELU
    d32 = H.make_conv_desc(B, one, one, l1.row0, l1.row0, 3 * 768, 512, co_pad, 1, 1, 0, 3 * 768, 512,
                           flags=base | _lib.SM_CONV_OUT_F32, acc_scale=1.0 / scale)
    d3 = H.make_conv_desc(B, one, one, l1.row0, l1.row0, 3 * 768, 512, co_pad, 1, 1, 0, 3 * 768, 3 * 512,
                          flags=base | _lib.SM_CONV_OUT_X3, acc_scale=1.0 / scale)
    y32 = torch.zeros(n0, 512, dtype=torch.float32, device=dev)
    y3 = torch.zeros(n0, 3 * 512, dtype=torch.float16, device=dev)
    H.conv2d(d32, cat3, wq, bias.to(dev), None, y32)
    H.conv2d(d3, cat3, wq, bias.to(dev), None, y3)
    r3 = torch.empty_like(y3)
    H.split3_f16(y32, r3, 512)
    torch.cuda.synchronize()
    assert torch.equal(y3, r3)
    ref = F.relu(F.conv2d(up.double(), wt.double(), bias.double()))
    got = y32.view(B, 24, 40, 512).permute(0, 3, 1, 2).cpu().double()
    assert float((got - ref).abs().max()) / float(ref.abs().max()) < 4e-6


def test_groupnorm_apply_x3_and_f32_statistics():
    """sm_gn_stats_f32_fix + sm_groupnorm_apply_x3 == F.group_norm + ReLU of the f32 rows (1e-5), written as f32 rows
    (in place) and as the next layer's split operand; the statistics are bit-reproducible"""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    B, C = 2, 256
    sizes = [(20, 33), (10, 17), (5, 9), (3, 5), (2, 3)]
    lv = H.Levels(B, sizes)
    xs = [torch.randn(B, C, h, w, generator=g) * 3 + 0.5 for h, w in sizes]
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    x = _rows(xs).to(dev)
    st = H.gn_stats_alloc(B * 5 * 32, dev)
    H.gn_stats_f32_fix(x, st, lv, C, 32)
    st2 = torch.full_like(st, 9)
    H.gn_stats_f32_fix(x, st2, lv, C, 32)
    torch.cuda.synchronize()
    assert torch.equal(st, st2)
    y3 = torch.empty(lv.rows, 3 * C, dtype=torch.float16, device=dev)
    H.groupnorm_apply_x3(x, gamma.to(dev), beta.to(dev), st, lv, C, 32, 1e-5, True, y_f32=x, y_split=y3)
    torch.cuda.synchronize()
    for l, (h, w) in enumerate(sizes):
        ref = F.relu(F.group_norm(xs[l].double(), 32, gamma.double(), beta.double(), 1e-5)).float()
        got = x[lv.row0[l]:lv.row0[l] + B * h * w].view(B, h, w, C).permute(0, 3, 1, 2).cpu()
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
        s3 = y3[lv.row0[l]:lv.row0[l] + B * h * w].float().cpu()
        rec = (s3[:, :C] + s3[:, C:2 * C]).view(B, h, w, C).permute(0, 3, 1, 2)
        torch.testing.assert_close(rec, got, rtol=2.0 ** -20, atol=1e-7)
        assert torch.equal(s3[:, :C], s3[:, 2 * C:])


@pytest.mark.parametrize("deform", [True, False])
def test_f32_conv_with_split_precision_contraction(deform):
    """sm_conv2d_f32 + SM_CONV_F16 (conv_f32.hip, X3): f32 rows in, operands split into binary16 halves in the loader,
    three f16 MFMAs per product.  FeatureAlign's shape (3x3, 256 -> 256, 4 deformable groups) against the oracle's
    deformable conv in float64 / torch's conv: 4e-6 of the output's largest value (the exact-f32 kernel: 1e-6)."""
    from oracle import ops as O
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    B, C, Co = 2, 256, 256
    sizes = [(21, 34), (9, 13)]
    lv = H.Levels(B, sizes)
    xs = [torch.randn(B, C, h, w, generator=g).abs() * 1.3 for h, w in sizes]
    offs = [torch.randn(B, 72, h, w, generator=g) * 1.5 for h, w in sizes]
    wt = torch.randn(Co, C, 3, 3, generator=g) * 0.03
    bias = torch.randn(Co, generator=g)
    scale = H.x3_weight_scale([wt])
    wq, co_pad = H.prep_conv_weight_f32(wt.to(dev) * scale, C)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, C, Co, co_pad, 3, 1, 1, C, Co, flags=_lib.SM_CONV_F16,
                         deform_groups=4 if deform else 0, acc_scale=1.0 / scale)
    y = torch.zeros(lv.rows, Co, dtype=torch.float32, device=dev)
    off_rows = _rows(offs).to(dev) if deform else None
    H.conv2d_f32(d, _rows(xs).to(dev), off_rows, wq, bias.to(dev), None, y)
    torch.cuda.synchronize()
    for l, (h, w) in enumerate(sizes):
        if deform:
            ref = O.deform_conv(xs[l].double(), offs[l].double(), wt.double(), 1, 1, 1, 4) + bias.double().view(1, -1, 1, 1)
        else:
            ref = F.conv2d(xs[l].double(), wt.double(), bias.double(), 1, 1)
        got = y[lv.row0[l]:lv.row0[l] + B * h * w].view(B, h, w, Co).permute(0, 3, 1, 2).cpu().double()
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        assert err < 4e-6, (deform, l, err)


@pytest.fixture(scope="module")
def head_case():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.manual_seed(0)
    B = 2
    sd = OM.init_state_dict(50, 0, calibrate=True)
    g = torch.Generator().manual_seed(1)
    sizes = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]
    feats = [torch.randn(B, 256, h, w, generator=g) * 2.0 for h, w in sizes]
    out = OM.head_forward(sd, feats)
    allc = torch.cat([c[0].reshape(-1) for c in out[0]]) - sd["bbox_head.fcos_cls.bias"][0]
    OM.calibrate_cls_bias(sd, allc, target=300)
    out = OM.head_forward(sd, feats)
    return dict(sd=sd, feats=feats, out=out, B=B, sizes=sizes)


def test_x3_head_matches_the_fp32_oracle_on_identical_features(head_case):
    """north_star: "outputs match the reference head on identical inputs ... mask logits within 1e-3".  The head-only
    x3 plan on the oracle's own f32 features: every head output within 1e-4 of its largest value, mask logits (basis .
    coefficients at the oracle's detections) within 1e-3 ABSOLUTE, the detections the SAME SET with identical labels."""
    from sipmask_amd.engine import SipMaskEngine
    c = head_case
    hsd = {k: v for k, v in c["sd"].items() if k.startswith("bbox_head.")}
    eng = SipMaskEngine.for_head(hsd, c["B"], c["sizes"], img_shape=(192, 256, 3), precision="head_x3")
    assert all(cv.mode in ("x3", "x3p", "x3w") for cv in eng.convs) and sum(cv.mode == "x3w" for cv in eng.convs) == 1
    # round 6: the 3x3 tower convs and fcos_cls + sip_cof read paired operands (three products on fragments read once)
    import sipmask_amd.engine as E
    paired = sorted(cv.name for cv in eng.convs if cv.mode == "x3p")
    assert paired == (["head.cls_cof", "head.reg_convs.3", "head.reg_ctr", "head.sip_mask_lat", "head.sip_mask_lat0",
                       "head.sip_mask_lat0.l1", "head.sip_mask_lat0.l2", "head.tower0", "head.tower1", "head.tower2"]
                      if E._X3_PAIRS else [])
    eng.load_pyramid([f.cuda() for f in c["feats"]])
    eng.run_head(with_post=True)
    torch.cuda.synchronize()
    cls, bb, ctr, cof, fm = eng.head_outputs()
    ocls, obb, octr, ocof, ofm = c["out"][:5]
    worst = {}
    for name, got, ref in (("cls", cls, ocls), ("bbox", bb, obb), ("ctr", ctr, octr), ("cof", cof, ocof)):
        e = max(float((a.cpu() - b).abs().max()) for a, b in zip(got, ref))
        m = max(float(b.abs().max()) for b in ref)
        worst[name] = e / m
        assert e <= 1e-4 * m, (name, e, m)
    assert float((fm.cpu() - ofm).abs().max()) <= 1e-4 * float(ofm.abs().max())
    res = eng.results()
    for b in range(c["B"]):
        r = OM.get_masks_single([x[b] for x in ocls], [x[b] for x in obb], [x[b] for x in octr], [x[b] for x in ocof], ofm[b],
                                (192, 256, 3), OM.DEFAULT_TEST_CFG)
        n = int(res["ndet"][b])
        assert n == r["det_bboxes"].shape[0] and n > 0
        # same detections in the same order unless two ranking keys are within rounding of each other: compare as sets
        got = sorted(zip(res["idxs_keep"][b, :n].cpu().tolist(), res["det_labels"][b, :n].cpu().tolist()))
        ref = sorted(zip(r["idxs_keep"].tolist(), r["det_labels"].tolist()))
        assert got == ref
        # mask logits at the oracle's detections
        keep = torch.as_tensor(r["idxs_keep"]).long()
        basis = fm[b].reshape(32, -1).t().cpu()
        obasis = ofm[b].reshape(32, -1).t()
        gcof = eng.sel["cofs"][b].cpu()[keep]
        ocf = torch.as_tensor(r["det_cofs"]).float()
        for q in range(4):
            lg = basis @ gcof[:, 32 * q:32 * q + 32].t()
            lr = obasis @ ocf[:, 32 * q:32 * q + 32].t()
            assert float((lg - lr).abs().max()) <= 1e-3, (b, q, float((lg - lr).abs().max()), float(lr.abs().max()))


def test_x3_plan_from_the_image_and_subbatch(head_case):
    """the whole plan (bf16 backbone + FPN, x3 head) runs, is bit-reproducible, and its HEAD reproduces the oracle head
    evaluated on the plan's own FPN features (what "identical inputs" means for a plan that starts at the image)."""
    from sipmask_amd.engine import SipMaskEngine
    sd = head_case["sd"]
    img = torch.randn(2, 3, 192, 256, generator=torch.Generator().manual_seed(3)).cuda()
    eng = SipMaskEngine(sd, 2, (192, 256), 50, precision="head_x3")
    r1 = {k: v.clone() for k, v in eng.run(img).items()}
    cc = eng.cls_cof.clone()
    r2 = eng.run(img)
    torch.cuda.synchronize()
    assert torch.equal(cc, eng.cls_cof) and all(torch.equal(r1[k], r2[k]) for k in r1)
    lv = eng.lv
    pyr = [eng.pyr[lv.row0[l]:lv.row0[l] + 2 * h * w].float().view(2, h, w, 256).permute(0, 3, 1, 2).cpu()
           for l, (h, w) in enumerate(lv.sizes)]
    out = OM.head_forward(sd, pyr)
    cls, bb, ctr, cof, fm = eng.head_outputs()
    for name, got, ref in (("cls", cls, out[0]), ("bbox", bb, out[1]), ("ctr", ctr, out[2]), ("cof", cof, out[3])):
        e = max(float((a.cpu() - b).abs().max()) for a, b in zip(got, ref))
        m = max(float(b.abs().max()) for b in ref)
        assert e <= 1e-4 * m, (name, e, m)
    assert float((fm.cpu() - out[4]).abs().max()) <= 1e-4 * float(out[4].abs().max())
