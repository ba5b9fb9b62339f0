"""FeatureAlign's deformable conv of the x3 head plan on the LDS-window kernel (csrc/deform_patch_x3.hip, round 5) and the
x3 build of the small-cout 3x3 kernel (csrc/conv3x3_smallco.hip, SM_CONV_F16).

The reference computes this conv in fp32 (M/mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:85-115,191-243 called from
M/mmdet/models/anchor_heads/sipmask_head.py:21-55); the checker is the oracle's restatement of that kernel evaluated in
float64 on the SAME f32 operands (oracle.ops.deform_conv, pinned to the reference's own .cu by tests/test_ref_pins.py).
Tolerance: 4e-6 of the output's largest value -- K = 2304 products of ~2^-21 each plus f32 accumulation order, the bound
the gather kernel it replaces is held to (tests/test_gpu_x3.py)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as O

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _rows(ts):
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts]).contiguous()


def _case(B, sizes, Co, off_scale, seed, G=4):
    g = torch.Generator().manual_seed(seed)
    C = 64 * G
    xs = [torch.randn(B, C, h, w, generator=g).abs() * 1.3 for h, w in sizes]
    offs = [torch.randn(B, G * 18, h, w, generator=g) * off_scale for h, w in sizes]
    wt = torch.randn(Co, C, 3, 3, generator=g) * 0.03
    bias = torch.randn(Co, generator=g)
    return xs, offs, wt, bias


def _window(B, sizes, xs, offs, wt, bias, flags=0, stats=False, G=4):
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    lv = H.Levels(B, sizes)
    C, Co = wt.shape[1], wt.shape[0]
    scale = H.x3_weight_scale([wt])
    wq, co_pad = H.prep_deform_weight_x3(wt.to(dev), scale, G)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, C, Co, co_pad, 3, 1, 1, C, Co, deform_groups=G,
                         flags=_lib.SM_CONV_F16 | _lib.SM_CONV_OUT_F32 | flags, acc_scale=1.0 / scale)
    assert H.deform_conv2d_x3_supported(d)
    y = torch.full((lv.rows, Co), float("nan"), dtype=torch.float32, device=dev)
    st = H.gn_stats_alloc(B * len(sizes) * (Co // 8), dev).fill_(7) if stats else None
    H.deform_conv2d_x3(d, _rows(xs).to(dev), _rows(offs).to(dev), wq, None if bias is None else bias.to(dev), y, st)
    torch.cuda.synchronize()
    return y, st, lv, d


def _gather(B, sizes, xs, offs, wt, bias, G=4):
    """the kernel it replaces: conv_f32.hip's gather loader with the split-precision contraction"""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    lv = H.Levels(B, sizes)
    C, Co = wt.shape[1], wt.shape[0]
    scale = H.x3_weight_scale([wt])
    wq, co_pad = H.prep_conv_weight_f32(wt.to(dev) * scale, C)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, C, Co, co_pad, 3, 1, 1, C, Co, flags=_lib.SM_CONV_F16,
                         deform_groups=G, acc_scale=1.0 / scale)
    y = torch.zeros(lv.rows, Co, dtype=torch.float32, device=dev)
    H.conv2d_f32(d, _rows(xs).to(dev), _rows(offs).to(dev), wq, bias.to(dev), None, y)
    torch.cuda.synchronize()
    return y


# levels that exercise every tile rule: 40 x 72 and 19 x 45 end in a strip of 32 x 8 column tiles, 10 x 23 keeps row tiles
# on its 23-column remainder, 33 x 8 is column tiles only, 2 x 3 is one mostly empty row tile
SIZES = [(40, 72), (19, 45), (10, 23), (33, 8), (2, 3)]


@pytest.mark.parametrize("off_scale", [0.0, 0.7, 2.0, 6.0])
def test_deform_x3_window_vs_float64_oracle(off_scale):
    """off_scale 0.7: every corner inside the LDS window (fast path only); 2.0: a far tap here and there between near ones
    (operands blended ahead by the previous step must be dropped); 6.0: nearly every wave gathers from global memory"""
    B, Co = 2, 256
    xs, offs, wt, bias = _case(B, SIZES, Co, off_scale, 11)
    y, _, lv, _ = _window(B, SIZES, xs, offs, wt, bias)
    assert torch.isfinite(y).all()
    old = _gather(B, SIZES, xs, offs, wt, bias)
    for l, (h, w) in enumerate(SIZES):
        ref = O.deform_conv(xs[l].double(), offs[l].double(), wt.double(), 1, 1, 1, 4) + bias.double().view(1, -1, 1, 1)
        got = y[lv.row0[l]:lv.row0[l] + B * h * w].view(B, h, w, Co).permute(0, 3, 1, 2).cpu().double()
        m = float(ref.abs().max())
        assert float((got - ref).abs().max()) < 4e-6 * m, (l, float((got - ref).abs().max()) / m)
        prev = old[lv.row0[l]:lv.row0[l] + B * h * w].view(B, h, w, Co).permute(0, 3, 1, 2).cpu().double()
        assert float((got - prev).abs().max()) < 4e-6 * m


def test_deform_x3_mixed_offsets_one_launch():
    """small offsets everywhere except a few positions that sample far outside the window (and outside the image), in a
    row tile and in a column tile: only the waves that own them leave the fast path"""
    B, Co = 1, 256
    sizes = [(24, 72)]
    xs, offs, wt, bias = _case(B, sizes, Co, 0.5, 9)
    o = offs[0].view(1, 4, 9, 2, 24, 72)
    o[0, 0, 4, :, 3, 5] = torch.tensor([9.5, -7.25])
    o[0, 2, 0, :, 11, 40] = torch.tensor([-30.0, 4.0])       # outside the image: zero sample
    o[0, 3, 8, :, 23, 71] = torch.tensor([-3.0, -3.0])       # exactly on the window edge (column tile)
    o[0, 1, 0, :, 0, 0] = torch.tensor([-0.5, -0.5])         # between the zero padding and pixel (0, 0)
    o[0, 1, 2, :, 16, 33] = torch.tensor([3.999, 3.999])
    o[0, 2, 5, :, 9, 68] = torch.tensor([12.0, -40.0])       # column tile, far
    y, _, lv, _ = _window(B, sizes, xs, offs, wt, bias)
    ref = O.deform_conv(xs[0].double(), offs[0].double(), wt.double(), 1, 1, 1, 4) + bias.double().view(1, -1, 1, 1)
    got = y.view(1, 24, 72, Co).permute(0, 3, 1, 2).cpu().double()
    assert float((got - ref).abs().max()) < 4e-6 * float(ref.abs().max())


def test_deform_x3_fused_gn_stats_relu_and_cout_tiles():
    """512 couts = two cout tiles; ReLU in the epilogue; the GroupNorm statistics are those of the PRE-activation output
    (FeatureAlign.forward normalises before the ReLU, sipmask_head.py:49-55) and agree with sm_gn_stats_f32_fix of it"""
    from sipmask_amd import hip_ops as H, _lib
    B, Co = 2, 512
    sizes = [(13, 37), (7, 11)]
    xs, offs, wt, bias = _case(B, sizes, Co, 1.0, 21)
    y, st, lv, _ = _window(B, sizes, xs, offs, wt, bias, stats=True)
    yr, _, _, _ = _window(B, sizes, xs, offs, wt, bias, flags=_lib.SM_CONV_RELU)
    assert torch.equal(yr, y.clamp_min(0))
    st2 = H.gn_stats_alloc(B * len(sizes) * (Co // 8), _dev())
    H.gn_stats_f32_fix(y, st2, lv, Co, Co // 8)
    a, b = H.gn_stats_to_float(st).cpu(), H.gn_stats_to_float(st2).cpu()
    torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-3)
    a = a.view(B, len(sizes), Co // 8, 2)
    for l, (h, w) in enumerate(sizes):
        ref = O.deform_conv(xs[l].double(), offs[l].double(), wt.double(), 1, 1, 1, 4) + bias.double().view(1, -1, 1, 1)
        r8 = ref.reshape(B, Co // 8, 8 * h * w)
        torch.testing.assert_close(a[:, l, :, 0], r8.sum(-1), rtol=1e-5, atol=1e-3 * (h * w) ** 0.5)
        torch.testing.assert_close(a[:, l, :, 1], (r8 * r8).sum(-1), rtol=1e-5, atol=1e-3)


def test_x3_small_cout_kernel_vs_float64():
    """sm_conv3x3_smallco with SM_CONV_F16: the x3 plan's sip_mask_lat (512 -> 32, ReLU) and fcos_reg + centerness
    (256 -> 5 padded to 8, per-level Scale on 4) on split operands -- within 4e-6 of the float64 conv of the f32 operands,
    and equal to the implicit-GEMM x3 launch up to accumulation order"""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    B = 2
    for (ci, co, sizes, relu, nch) in [(512, 32, [(25, 40)], True, 0), (256, 5, [(20, 33), (10, 17), (5, 9)], False, 4)]:
        lv = H.Levels(B, sizes)
        xs = [torch.randn(B, ci, h, w, generator=g).abs() for h, w in sizes]
        wt = torch.randn(co, ci, 3, 3, generator=g) * (0.5 / (ci * 9) ** 0.5)
        bias = torch.randn(co, generator=g)
        cs = (co + 7) // 8 * 8
        wt8 = torch.cat([wt, torch.zeros(cs - co, ci, 3, 3)], 0)
        b8 = torch.cat([bias, torch.zeros(cs - co)], 0)
        lscale = [1.0 + 0.25 * l for l in range(len(sizes))]
        x3 = torch.empty(lv.rows, 3 * ci, dtype=torch.float16, device=dev)
        H.split3_f16(_rows(xs).to(dev), x3)
        scale = H.x3_weight_scale([wt8])
        wf = H.prep_conv_weight_smallco(wt8.to(dev), x3_scale=scale)
        assert wf.dtype == torch.float16 and tuple(wf.shape) == (3 * ci // 32, 9, 2, 64, 8)
        d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 3 * ci, cs, 32, 3, 1, 1, 3 * ci, cs,
                             flags=_lib.SM_CONV_F16 | _lib.SM_CONV_OUT_F32 | (_lib.SM_CONV_RELU if relu else 0), scale_nch=nch,
                             level_scale=lscale, acc_scale=1.0 / scale)
        assert H.conv3x3_smallco_supported(d)
        y = torch.full((lv.rows, cs), float("nan"), dtype=torch.float32, device=dev)
        H.conv3x3_smallco(d, x3, wf, b8.to(dev), y)
        torch.cuda.synchronize()
        for l, (h, w) in enumerate(sizes):
            ref = F.conv2d(xs[l].double(), wt.double(), bias.double(), 1, 1)
            if nch:
                ref[:, :nch] *= lscale[l]
            if relu:
                ref = ref.clamp_min(0)
            got = y[lv.row0[l]:lv.row0[l] + B * h * w, :co].view(B, h, w, co).permute(0, 3, 1, 2).cpu().double()
            err = float((got - ref).abs().max()) / float(ref.abs().max())
            assert err < 4e-6, (ci, co, l, err)
    d.flags &= ~_lib.SM_CONV_OUT_F32            # binary16 operands need f32 output
    assert not H.conv3x3_smallco_supported(d)


def test_x3_plan_uses_the_window_and_small_cout_kernels():
    """the x3 head plan routes FeatureAlign to the window kernel (statistics fused: the separate pass is a no-op) and
    sip_mask_lat / reg_ctr to the small-cout kernel; pinning the gather kernel gives the same head outputs up to
    accumulation order"""
    from oracle import model as OM
    from sipmask_amd import engine as E
    sd = OM.init_state_dict(50, 0, calibrate=True)
    hsd = {k: v for k, v in sd.items() if k.startswith("bbox_head.")}
    sizes = [(24, 40), (12, 20), (6, 10), (3, 5), (2, 3)]
    g = torch.Generator().manual_seed(2)
    feats = [(torch.randn(2, 256, h, w, generator=g) * 2.0).cuda() for h, w in sizes]
    eng = E.SipMaskEngine.for_head(hsd, 2, sizes, img_shape=(192, 320, 3), precision="head_x3")
    fa = eng._fa_conv
    assert fa.kernel == "window" and fa.mode == "x3w" and fa.fused_stats
    by_name = {c.name: c for c in eng.convs}
    assert by_name["head.sip_mask_lat"].smallco and by_name["head.reg_ctr"].smallco
    eng.load_pyramid(feats)
    eng.run_head(with_post=False)
    torch.cuda.synchronize()
    a = [eng.cls_cof.clone(), eng.reg_out.clone(), eng.basis_lo.clone()]
    fa.pick("gather")
    assert fa.mode == "f32x3" and not fa.fused_stats
    eng.run_head(with_post=False)
    torch.cuda.synchronize()
    b = [eng.cls_cof, eng.reg_out, eng.basis_lo]
    for u, v in zip(a, b):
        assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max())
