"""The LDS-patch deformable conv (csrc/deform_patch.hip) against the gather loader it replaces (conv_igemm.hip DEFORM, behind
SM_CONV_DBG_DEFORM_GATHER) and against the oracle restatement of deform_conv_cuda_kernel.cu:85-115,191-243 -- small offsets
(every corner inside the LDS window), large offsets (the per-wave global fallback), both in one launch, tiles that hang
over the image, fused GroupNorm statistics and the bf16 epilogue."""
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _bf(t):
    return t.to(torch.bfloat16).float()


def _run(B, sizes, C, Co, G, off_rows, x_rows, w, flags, gather, stats=False, bias=None, out_dtype=torch.float32):
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    lv = H.Levels(B, sizes)
    wq, co_pad = H.prep_conv_weight(w.to(dev))
    fl = flags | (_lib.SM_CONV_DBG_DEFORM_GATHER if gather else 0)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, C, Co, co_pad, 3, 1, 1, C, Co, flags=fl, deform_groups=G)
    y = torch.full((lv.rows, Co), float("nan"), dtype=out_dtype, device=dev)
    if stats:
        st = H.gn_stats_alloc(B * len(sizes) * (Co // 8), dev)
        H.conv2d_gn_stats(d, x_rows, off_rows, wq, bias, None, y, st)
        torch.cuda.synchronize()
        return y, st
    H.deform_conv2d(d, x_rows, off_rows, wq, bias, y)
    torch.cuda.synchronize()
    return y, None


def _close_to_gather(got, old):
    """Same samples, K summed in (group, tap, channel) instead of (tap, channel) order: accumulation-order noise everywhere,
    plus the rare sample (measured 165 of 600 k outputs touched) whose f32 blend lands on the other side of a bf16 rounding
    boundary because hipcc contracts the corner-weight products differently in the two kernels (1 f32 ulp)."""
    torch.testing.assert_close(got, old, rtol=2e-3, atol=2e-3)
    far = (got - old).abs() > 2e-4 + 1e-4 * old.abs()
    assert float(far.float().mean()) < 2e-3, float(far.float().mean())


def _inputs(B, sizes, C, Co, G, off_scale, seed):
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(seed)
    xs = [_bf(torch.randn(B, C, h, w, generator=g)) for h, w in sizes]
    offs = [torch.randn(B, G * 18, h, w, generator=g) * off_scale for h, w in sizes]
    wt = _bf(torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    lv = H.Levels(B, sizes)
    x_rows = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, C) for x in xs]).to(torch.bfloat16).to(dev).contiguous()
    off_rows = torch.cat([o.permute(0, 2, 3, 1).reshape(-1, G * 18) for o in offs]).to(dev).contiguous()
    assert x_rows.shape[0] == lv.rows
    return xs, offs, wt, x_rows, off_rows, lv


@pytest.mark.parametrize("off_scale", [0.0, 0.7, 6.0])
def test_deform_patch_vs_gather_and_oracle(off_scale):
    """five levels incl. tiles that are mostly outside the image; off_scale 0.7: every corner in the window (fast path only),
    6.0: nearly every wave takes the global fallback"""
    from sipmask_amd import _lib
    B, C, Co, G = 2, 256, 256, 4
    sizes = [(19, 45), (10, 23), (5, 12), (3, 6), (2, 3)]
    xs, offs, wt, x_rows, off_rows, lv = _inputs(B, sizes, C, Co, G, off_scale, 5)
    got, _ = _run(B, sizes, C, Co, G, off_rows, x_rows, wt, _lib.SM_CONV_OUT_F32, gather=False)
    old, _ = _run(B, sizes, C, Co, G, off_rows, x_rows, wt, _lib.SM_CONV_OUT_F32, gather=True)
    assert torch.isfinite(got).all()
    _close_to_gather(got, old)
    for l, (h, w) in enumerate(sizes):
        ref = O.deform_conv(xs[l], offs[l], wt, 1, 1, 1, G, col_round=lambda t: t.to(torch.bfloat16).float())
        out = got[lv.row0[l]:lv.row0[l] + B * h * w].view(B, h, w, Co).permute(0, 3, 1, 2).cpu()
        torch.testing.assert_close(out, ref, rtol=2e-3, atol=2e-3)


def test_deform_patch_mixed_offsets_one_launch():
    """small offsets everywhere except a few positions that sample far outside the window (and outside the image): only the
    waves that own them leave the fast path, the result is the gather loader's either way"""
    from sipmask_amd import _lib
    B, C, Co, G = 1, 256, 256, 4
    sizes = [(24, 70)]
    xs, offs, wt, x_rows, off_rows, lv = _inputs(B, sizes, C, Co, G, 0.5, 9)
    o = off_rows.view(24, 70, G, 9, 2)
    o[3, 5, 0, 4] = torch.tensor([9.5, -7.25])
    o[11, 40, 2, 0] = torch.tensor([-30.0, 4.0])        # outside the image: zero sample
    o[23, 69, 3, 8] = torch.tensor([-3.0, -3.0])        # exactly on the window edge
    o[0, 0, 1, 0] = torch.tensor([-0.5, -0.5])          # between the zero padding and pixel (0, 0)
    o[16, 33, 1, 2] = torch.tensor([3.999, 3.999])
    got, _ = _run(B, sizes, C, Co, G, off_rows, x_rows, wt, _lib.SM_CONV_OUT_F32, gather=False)
    old, _ = _run(B, sizes, C, Co, G, off_rows, x_rows, wt, _lib.SM_CONV_OUT_F32, gather=True)
    _close_to_gather(got, old)
    offs_t = off_rows.view(1, 24, 70, G * 18).permute(0, 3, 1, 2).cpu()
    ref = O.deform_conv(xs[0], offs_t, wt, 1, 1, 1, G, col_round=lambda t: t.to(torch.bfloat16).float())
    torch.testing.assert_close(got.view(1, 24, 70, Co).permute(0, 3, 1, 2).cpu(), ref, rtol=2e-3, atol=2e-3)


def test_deform_patch_gn_stats_bias_relu_bf16():
    """FeatureAlign's launch: bf16 rows out, GroupNorm statistics of the (pre-activation) output fused, 512 couts = 2 tiles"""
    from sipmask_amd import _lib, hip_ops as H
    B, C, Co, G = 2, 256, 512, 4
    sizes = [(13, 37), (7, 11)]
    xs, offs, wt, x_rows, off_rows, lv = _inputs(B, sizes, C, Co, G, 1.0, 21)
    bias = torch.randn(Co, generator=torch.Generator().manual_seed(3)).to(_dev())
    got, st = _run(B, sizes, C, Co, G, off_rows, x_rows, wt, _lib.SM_CONV_RELU, False, stats=True, bias=bias, out_dtype=torch.bfloat16)
    old, st_old = _run(B, sizes, C, Co, G, off_rows, x_rows, wt, _lib.SM_CONV_RELU, True, stats=True, bias=bias, out_dtype=torch.bfloat16)
    torch.testing.assert_close(got.float(), old.float(), rtol=2 ** -7, atol=2e-3)
    st, st_old = H.gn_stats_to_float(st), H.gn_stats_to_float(st_old)
    torch.testing.assert_close(st, st_old, rtol=2e-4, atol=5e-2)
    st = st.view(B, len(sizes), Co // 8, 2).cpu()
    for l, (h, w) in enumerate(sizes):
        ref = O.deform_conv(xs[l], offs[l], wt, 1, 1, 1, G, col_round=lambda t: t.to(torch.bfloat16).float()).double()
        ref = ref + bias.cpu().double().view(1, -1, 1, 1)
        rs = ref.view(B, Co // 8, 8 * h * w)
        torch.testing.assert_close(st[:, l, :, 0], rs.sum(-1), rtol=1e-3, atol=5e-2)
        torch.testing.assert_close(st[:, l, :, 1], (rs * rs).sum(-1), rtol=1e-3, atol=5e-2)
        out = got[lv.row0[l]:lv.row0[l] + B * h * w].float().view(B, h, w, Co).permute(0, 3, 1, 2).cpu()
        torch.testing.assert_close(out, ref.float().clamp_min(0), rtol=2 ** -7, atol=3e-3)


def test_deform_patch_not_taken_for_other_shapes():
    """64 channels in 4 groups (16 per group) is not the patch kernel's shape: same result with and without the A/B flag,
    bit for bit, because both go through the gather loader"""
    from sipmask_amd import _lib
    B, C, Co, G = 1, 64, 40, 4
    sizes = [(9, 9)]
    xs, offs, wt, x_rows, off_rows, lv = _inputs(B, sizes, C, Co, G, 1.0, 2)
    a, _ = _run(B, sizes, C, Co, G, off_rows, x_rows, wt, _lib.SM_CONV_OUT_F32, gather=False)
    b, _ = _run(B, sizes, C, Co, G, off_rows, x_rows, wt, _lib.SM_CONV_OUT_F32, gather=True)
    assert torch.equal(a, b)
