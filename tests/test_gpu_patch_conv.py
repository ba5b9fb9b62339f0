"""GPU parity of the patch-resident 3x3 conv kernel (csrc/conv3x3_patch.hip, sm_conv3x3_patch) against torch fp32
F.conv2d on bf16-representable inputs: f32 outputs within accumulation order (rtol 1e-4 / atol 2e-4), bf16 outputs
plus one rounding; multi-level launches (tiles never straddle an image), ragged widths (pad columns are dummy outputs),
per-level Scale, ReLU, fused GroupNorm statistics, the group dimension (cls + reg towers in one launch)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _rows(ts):
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts])


@pytest.mark.parametrize("cfg", [
    # B, Cin, Cout, sizes
    (2, 256, 256, [(25, 42), (13, 21), (7, 11), (4, 6), (2, 3)]),     # tower-like pyramid
    (1, 64, 208, [(19, 37)]),                                          # cout tail inside a 256 tile, odd width
    (2, 128, 512, [(9, 250)]),                                         # two cout tiles, widest supported rows
    (1, 256, 256, [(100, 168)]),                                       # FPN level-0 geometry (67 tiles)
])
@pytest.mark.parametrize("shape_flag", [0, 0x4000])   # the planner's launch shape / uniform 256-position tiles (SM_CONV_DBG_PATCH_UNIFORM)
def test_patch_conv_vs_torch(cfg, shape_flag):
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    B, Ci, Co, sizes = cfg
    g = torch.Generator().manual_seed(Ci + Co + len(sizes))
    lv = H.Levels(B, sizes)
    xs = [_bf(torch.randn(B, Ci, h, w, generator=g)) for h, w in sizes]
    w = _bf(torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5)
    bias = torch.randn(Co, generator=g)
    x = _rows(xs).to(torch.bfloat16).to(dev)
    wq, co_pad = H.prep_conv_weight_patch(w.to(dev))
    scales = [1.0 + 0.25 * l for l in range(len(sizes))]
    for out_f32, relu in ((True, False), (False, True)):
        flags = (_lib.SM_CONV_OUT_F32 if out_f32 else 0) | (_lib.SM_CONV_RELU if relu else 0) | shape_flag
        d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, Ci, Co, co_pad, 3, 1, 1, Ci, Co, flags=flags,
                             scale_nch=4, level_scale=scales)
        assert H.conv3x3_patch_supported(d)
        y = torch.full((lv.rows, Co), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
        H.conv3x3_patch(d, x, wq, bias.to(dev), y)
        torch.cuda.synchronize()
        for l, (h, wd) in enumerate(sizes):
            ref = F.conv2d(xs[l], w, bias, 1, 1)
            ref[:, :4] *= scales[l]
            if relu:
                ref = F.relu(ref)
            got = y[lv.row0[l]:lv.row0[l] + B * h * wd].float().view(B, h, wd, Co).permute(0, 3, 1, 2).cpu()
            if out_f32:
                torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-4)
            else:
                torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("shared_x", [True, False])
def test_patch_conv_grouped_with_groupnorm_statistics(shared_x):
    """two problem instances (the cls and reg tower convs of one depth) in ONE launch + fused GN statistics"""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(3 + shared_x)
    B, C, G = 2, 256, 2
    sizes = [(20, 33), (10, 17), (5, 9)]
    lv = H.Levels(B, sizes)
    xs = [[_bf(torch.randn(B, C, h, w, generator=g)) for h, w in sizes] for _ in range(1 if shared_x else G)]
    ws = [_bf(torch.randn(C, C, 3, 3, generator=g) / 48) for _ in range(G)]
    x = torch.cat([_rows(t) for t in xs]).to(torch.bfloat16).to(dev)
    packed = [H.prep_conv_weight_patch(w.to(dev))[0] for w in ws]
    wq = torch.stack(packed).contiguous()
    y = torch.zeros(G * lv.rows, C, dtype=torch.bfloat16, device=dev)
    S = 2 * B * len(sizes) * (C // 8)
    stats = torch.full((G * S,), 7, dtype=torch.int64, device=dev)      # zeroed by the call
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, C, C, 256, 3, 1, 1, C, C, ngroups=G,
                         x_group_rows=0 if shared_x else lv.rows, y_group_rows=lv.rows, w_group_stride=packed[0].numel(),
                         bias_group_stride=0, gn_group_stride=S)
    H.conv3x3_patch(d, x, wq, None, y, stats)
    torch.cuda.synchronize()
    for gi in range(G):
        src = xs[0] if shared_x else xs[gi]
        st = H.gn_stats_to_float(stats[gi * S:(gi + 1) * S].view(B, len(sizes), C // 8, 2).cpu()).float()
        for l, (h, wd) in enumerate(sizes):
            ref = F.conv2d(src[l], ws[gi], None, 1, 1)
            got = y[gi * lv.rows + lv.row0[l]: gi * lv.rows + lv.row0[l] + B * h * wd].float().view(B, h, wd, C).permute(0, 3, 1, 2).cpu()
            torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=2e-3)
            r8 = ref.view(B, C // 8, 8, h * wd)
            torch.testing.assert_close(st[:, l, :, 0], r8.sum((2, 3)), rtol=2e-3, atol=0.5)
            torch.testing.assert_close(st[:, l, :, 1], (r8 * r8).sum((2, 3)), rtol=2e-3, atol=0.5)


@pytest.mark.parametrize("batch,groups", [(2, 2), (4, 1), (2, 1)])
def test_patch_conv_mixed_tile_launch_at_head_shape(batch, groups):
    """the BASELINE head pyramid (5 levels of 800x1344): launches the planner cuts into 256-position tiles + 128- / 192-
    position finishing tiles (sm_conv3x3_patch_plan), grouped and not, with fused GN statistics; reference = torch f32
    conv on the device, and the uniform launch (SM_CONV_DBG_PATCH_UNIFORM) must give bit-identical outputs (same K
    order per output element)."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(11 + batch + groups)
    C = 256
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    lv = H.Levels(batch, sizes)
    xs = [_bf(torch.randn(batch, C, h, w, generator=g)).to(dev) for h, w in sizes]
    ws = [_bf(torch.randn(C, C, 3, 3, generator=g) / 48).to(dev) for _ in range(groups)]
    x = _rows(xs).to(torch.bfloat16)
    packed = [H.prep_conv_weight_patch(w)[0] for w in ws]
    wq = torch.stack(packed).contiguous()
    S = 2 * batch * len(sizes) * (C // 8)
    outs = {}
    for flag in (0, 0x4000):
        y = torch.zeros(groups * lv.rows, C, dtype=torch.bfloat16, device=dev)
        stats = torch.full((groups * S,), 7, dtype=torch.int64, device=dev)
        d = H.make_conv_desc(batch, sizes, sizes, lv.row0, lv.row0, C, C, 256, 3, 1, 1, C, C, flags=flag, ngroups=groups,
                             x_group_rows=0, y_group_rows=lv.rows, w_group_stride=packed[0].numel(), bias_group_stride=0,
                             gn_group_stride=S)
        pl = H.conv3x3_patch_plan(d)
        if not (flag & 0x4000):
            assert pl["small"] > 0, pl            # these shapes are the ones the mixed launch exists for
        else:
            assert pl["small"] == 0, pl
        H.conv3x3_patch(d, x, wq, None, y, stats)
        torch.cuda.synchronize()
        # run to run: the statistics are fixed-point sums (integer atomics), so a second launch reproduces them bit for bit
        again = torch.full_like(stats, 3)
        H.conv3x3_patch(d, x, wq, None, y, again)
        torch.cuda.synchronize()
        assert torch.equal(stats, again)
        outs[flag] = (y, stats)
    assert torch.equal(outs[0][0], outs[0x4000][0])
    # another tile cut, the SAME statistics: a contribution is the sum over 32 consecutive padded-flat positions x 8 couts
    # whatever tile they fall into, and the sums of contributions are integer
    assert torch.equal(outs[0][1], outs[0x4000][1])
    y, stats = outs[0]
    for gi in range(groups):
        st = H.gn_stats_to_float(stats[gi * S:(gi + 1) * S].view(batch, len(sizes), C // 8, 2)).float()
        for l, (h, wd) in enumerate(sizes):
            ref = F.conv2d(xs[l], ws[gi], None, 1, 1)
            got = y[gi * lv.rows + lv.row0[l]: gi * lv.rows + lv.row0[l] + batch * h * wd].float().view(batch, h, wd, C).permute(0, 3, 1, 2)
            torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=2e-3)
            r8 = ref.reshape(batch, C // 8, 8, h * wd)
            torch.testing.assert_close(st[:, l, :, 0], r8.sum((2, 3)), rtol=2e-3, atol=1.0)
            torch.testing.assert_close(st[:, l, :, 1], (r8 * r8).sum((2, 3)), rtol=2e-3, atol=1.0)


@pytest.mark.parametrize("cfg", [
    # B, Cin, Cout, sizes: the two convs the 32-cout tile exists for, at reduced sizes, and odd geometries
    (2, 512, 32, [(25, 42)]),                                          # sip_mask_lat: 512 -> 32 on the stride-8 grid
    (2, 256, 8, [(25, 42), (13, 21), (7, 11), (4, 6), (2, 3)]),         # fcos_reg + centerness (+3 zero channels) over the pyramid
    (1, 64, 24, [(19, 37), (9, 250)]),                                  # cout tail inside the 32 tile, odd / widest rows
])
def test_patch_conv_small_cout_tile_vs_torch(cfg):
    """round 4: sm_conv3x3_patch with weights padded to 32 cout rows runs its 32-cout x 256-position tile (one MFMA tile per
    wave): same contract as the 256-cout tile -- f32 outputs within accumulation order, bf16 plus one rounding, per-level
    Scale on the first channels, ReLU, multi-level launches -- and the same bits as the implicit-GEMM kernel's K order is
    not required (patch order = chunk, tap, channel)."""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    B, Ci, Co, sizes = cfg
    g = torch.Generator().manual_seed(Ci + Co + len(sizes))
    lv = H.Levels(B, sizes)
    xs = [_bf(torch.randn(B, Ci, h, w, generator=g)) for h, w in sizes]
    w = _bf(torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5)
    bias = torch.randn(Co, generator=g)
    x = _rows(xs).to(torch.bfloat16).to(dev)
    assert H.patch_cout_pad(Co) == 32
    wq, co_pad = H.prep_conv_weight_patch(w.to(dev), 32)
    assert co_pad == 32 and wq.shape == (32, 9 * Ci)
    scales = [1.0 + 0.25 * l for l in range(len(sizes))]
    for out_f32, relu in ((True, False), (False, True)):
        flags = (_lib.SM_CONV_OUT_F32 if out_f32 else 0) | (_lib.SM_CONV_RELU if relu else 0)
        d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, Ci, Co, co_pad, 3, 1, 1, Ci, Co, flags=flags,
                             scale_nch=4, level_scale=scales)
        assert H.conv3x3_patch_supported(d)
        y = torch.full((lv.rows, Co), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
        H.conv3x3_patch(d, x, wq, bias.to(dev), y)
        torch.cuda.synchronize()
        for l, (h, wd) in enumerate(sizes):
            ref = F.conv2d(xs[l], w, bias, 1, 1)
            ref[:, :4] *= scales[l]
            if relu:
                ref = F.relu(ref)
            got = y[lv.row0[l]:lv.row0[l] + B * h * wd].float().view(B, h, wd, Co).permute(0, 3, 1, 2).cpu()
            if out_f32:
                torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-4)
            else:
                torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=2e-3)
    # grouped / per-level launches keep the 256-cout tile: the library refuses them on 32-row weights
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, Ci, Co, 32, 3, 1, 1, Ci, Co, ngroups=2, y_group_rows=lv.rows,
                         w_group_stride=wq.numel())
    assert not H.conv3x3_patch_supported(d)


@pytest.mark.parametrize("cfg", [
    (2, 512, 32, [(25, 42)], 512, 32, 0),                                      # sip_mask_lat
    (2, 256, 8, [(25, 42), (13, 21), (7, 11), (4, 6), (2, 3)], 256, 8, 0),     # fcos_reg + centerness over the pyramid
    (1, 64, 24, [(19, 37), (9, 250), (1, 1)], 96, 40, 8),                      # input / output channel slices, 1-pixel level
    (3, 32, 16, [(5, 33)], 32, 16, 0),                                         # a single channel slice, 33 columns (2 tiles)
    (1, 96, 32, [(3, 64)], 96, 32, 0),                                         # odd slice count (3), odd row count
])
def test_conv3x3_smallco_vs_torch(cfg):
    """round 4: sm_conv3x3_smallco -- the 3x3 convs with 8..32 output channels (sip_mask_lat, fcos_reg + centerness) on their
    own kernel (one wave per 2 x 32-position tile, three channel slices in flight, weights as MFMA fragments from L2): the
    contract of sm_conv2d's epilogue (bias, per-level Scale on the first channels, ReLU / ReLU on those channels only, f32 or
    bf16 rows, channel slices of wider row tensors) against torch f32 on the same bf16 operands."""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    B, Ci, Co, sizes, in_cs, out_cs, out_coff = cfg
    g = torch.Generator().manual_seed(Ci + Co + len(sizes))
    lv = H.Levels(B, sizes)
    xs = [_bf(torch.randn(B, Ci, h, w, generator=g)) for h, w in sizes]
    w = _bf(torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5)
    bias = torch.randn(Co, generator=g)
    x = torch.full((lv.rows, in_cs), float("nan"), dtype=torch.bfloat16)       # channels beyond cin must never be read
    x[:, :Ci] = _rows(xs).to(torch.bfloat16)
    x = x.to(dev)
    wq = H.prep_conv_weight_smallco(w.to(dev))
    assert tuple(wq.shape) == (Ci // 32, 9, 2, 64, 8)
    scales = [1.0 + 0.25 * l for l in range(len(sizes))]
    for out_f32, relu in ((True, 0), (False, _lib.SM_CONV_RELU), (True, _lib.SM_CONV_RELU_NCH)):
        flags = (_lib.SM_CONV_OUT_F32 if out_f32 else 0) | relu
        d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, Ci, Co, 32, 3, 1, 1, in_cs, out_cs, out_coff, flags=flags,
                             scale_nch=4, level_scale=scales)
        assert H.conv3x3_smallco_supported(d)
        assert H.conv3x3_smallco_tiles(d) == sum(B * -(-h // 2) * -(-wd // 32) for h, wd in sizes)
        y = torch.full((lv.rows, out_cs), -7.0, dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
        H.conv3x3_smallco(d, x, wq, bias.to(dev), y)
        torch.cuda.synchronize()
        for l, (h, wd) in enumerate(sizes):
            ref = F.conv2d(xs[l], w, bias, 1, 1)
            ref[:, :4] *= scales[l]
            if relu == _lib.SM_CONV_RELU:
                ref = F.relu(ref)
            elif relu:
                ref[:, :4] = F.relu(ref[:, :4])
            blk = y[lv.row0[l]:lv.row0[l] + B * h * wd].float().cpu()
            got = blk[:, out_coff:out_coff + Co].view(B, h, wd, Co).permute(0, 3, 1, 2)
            if out_f32:
                torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-4)
            else:
                torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=2e-3)
            rest = torch.cat([blk[:, :out_coff], blk[:, out_coff + Co:]], 1)     # neighbouring channel slices untouched
            assert rest.numel() == 0 or bool((rest == -7.0).all())
    # what the kernel does not do is refused, not approximated
    for kw in (dict(flags=_lib.SM_CONV_RES_ADD), dict(ngroups=2, y_group_rows=lv.rows, w_group_stride=wq.numel())):
        d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, Ci, Co, 32, 3, 1, 1, in_cs, out_cs, out_coff, **kw)
        assert not H.conv3x3_smallco_supported(d)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, Ci, 64, 64, 3, 1, 1, in_cs, 64)
    assert not H.conv3x3_smallco_supported(d)


@pytest.mark.parametrize("cfg", [(2, 256, 256, (50, 84)), (1, 512, 512, (25, 42)), (2, 64, 128, (19, 37))])
def test_patch_conv_cout128_tile_vs_torch(cfg):
    """round 4: sm_conv_desc.patch_cout_tile = 128 -- 128-cout x 256-position tiles for the 3x3 convs of ResNet layer3 / layer4
    (too few positions for 256-cout tiles): f32 outputs within accumulation order of torch, bf16 + ReLU plus one rounding, and
    BIT-IDENTICAL to the 256-cout tile of the same kernel (same K order per output element)."""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    B, Ci, Co, (h, wd) = cfg
    g = torch.Generator().manual_seed(Ci + Co + h)
    xs = _bf(torch.randn(B, Ci, h, wd, generator=g))
    w = _bf(torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5)
    bias = torch.randn(Co, generator=g)
    x = _rows([xs]).to(torch.bfloat16).to(dev)
    wq, co_pad = H.prep_conv_weight_patch(w.to(dev), Co)
    assert co_pad == Co
    rows = B * h * wd
    for out_f32, relu in ((True, False), (False, True)):
        flags = (_lib.SM_CONV_OUT_F32 if out_f32 else 0) | (_lib.SM_CONV_RELU if relu else 0)
        d = H.make_conv_desc(B, [(h, wd)], [(h, wd)], [0], [0], Ci, Co, co_pad, 3, 1, 1, Ci, Co, flags=flags, patch_cout_tile=128)
        assert H.conv3x3_patch_supported(d) and H.conv3x3_patch_plan(d)["small"] == 0
        y = torch.full((rows, Co), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
        H.conv3x3_patch(d, x, wq, bias.to(dev), y)
        torch.cuda.synchronize()
        ref = F.conv2d(xs, w, bias, 1, 1)
        if relu:
            ref = F.relu(ref)
        got = y.float().view(B, h, wd, Co).permute(0, 3, 1, 2).cpu()
        if out_f32:
            torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-4)
        else:
            torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=2e-3)
        if Co % 256 == 0:
            wq256, cp = H.prep_conv_weight_patch(w.to(dev))
            d256 = H.make_conv_desc(B, [(h, wd)], [(h, wd)], [0], [0], Ci, Co, cp, 3, 1, 1, Ci, Co, flags=flags | 0x4000)
            y256 = torch.empty_like(y)
            H.conv3x3_patch(d256, x, wq256, bias.to(dev), y256)
            torch.cuda.synchronize()
            assert torch.equal(y, y256)
    bad = H.make_conv_desc(B, [(h, wd)], [(h, wd)], [0], [0], Ci, Co, co_pad, 3, 1, 1, Ci, Co, patch_cout_tile=64)
    assert not H.conv3x3_patch_supported(bad)
