"""GPU parity tests: every HIP kernel (through the C ABI) against the CPU oracle / a plain
torch-fp32 reference on identical seeded inputs.  Tolerances are stated per test:
  * integer / index outputs (NMS keep, top-k, labels, crop cells, masks away from the threshold): bit-exact
  * bf16-input MFMA convs: inputs are pre-rounded to bf16 so the only error is f32 accumulation order
    (|err| <= 2e-3 * scale) plus one bf16 rounding of the output where the output is bf16 (rel 2^-8)
  * f32 elementwise kernels: 1e-5 relative (expf/logf ulp differences)
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ops as O  # noqa: E402
from sipmask_amd import _lib as L  # noqa: E402


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _run_conv(x, w, bias, stride, pad, relu=False, out_f32=False, residual=None, in_relu=False, extra_flags=0, want_plan=None):
    """x [B,C,H,W] f32 (bf16-representable), w [Co,Ci,k,k]; returns NCHW f32 on cpu."""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    B, C, Hh, Ww = x.shape
    Co, _, k, _ = w.shape
    cpad = (C + 7) // 8 * 8
    xh = torch.empty(B * Hh * Ww, cpad, dtype=torch.bfloat16, device=dev)
    H.nchw_to_nhwc_bf16(x.to(dev).contiguous(), xh, cpad)
    wq, co_pad = H.prep_conv_weight(w.to(dev), cpad)
    Ho, Wo = (Hh + 2 * pad - k) // stride + 1, (Ww + 2 * pad - k) // stride + 1
    flags = (_lib.SM_CONV_RELU if relu else 0) | (_lib.SM_CONV_OUT_F32 if out_f32 else 0)
    flags |= _lib.SM_CONV_IN_RELU if in_relu else 0
    flags |= extra_flags
    res = None
    if residual is not None:
        flags |= _lib.SM_CONV_RES_ADD
        res = residual.permute(0, 2, 3, 1).reshape(-1, Co).to(torch.bfloat16).to(dev).contiguous()
    y = torch.empty(B * Ho * Wo, Co, dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    d = H.make_conv_desc(B, [(Hh, Ww)], [(Ho, Wo)], [0], [0], cpad, Co, co_pad, k, stride, pad, cpad, Co,
                         flags=flags, res_cstride=Co)
    if want_plan is not None:
        want_plan.update(H.conv_plan(d))
    H.conv2d(d, xh, wq, None if bias is None else bias.to(dev), res, y)
    torch.cuda.synchronize()
    return y.float().view(B, Ho, Wo, Co).permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("cfg", [
    # (B, Cin, H, W, Cout, k, stride, pad)
    (2, 64, 17, 23, 128, 3, 1, 1),     # 128-tile, ragged M
    (1, 256, 13, 21, 256, 3, 1, 1),    # tower shape
    (2, 128, 20, 28, 64, 1, 1, 0),     # 64-tile, 1x1
    (2, 64, 21, 19, 5, 3, 1, 1),       # 32-tile, cout tail (reg+ctr)
    (2, 3, 37, 45, 64, 7, 2, 3),       # stem: cin padded 3->8, K padded to 448
    (1, 512, 16, 12, 1024, 1, 2, 0),   # strided 1x1 (downsample)
    (2, 256, 9, 11, 208, 3, 1, 1),     # cls+cof fused width
    (1, 768, 10, 14, 512, 1, 1, 0),    # cin/8 not a power of two
    (1, 256, 25, 42, 256, 3, 2, 1),    # P6: stride-2 3x3
])
def test_conv_igemm_vs_torch(cfg):
    B, Ci, Hh, Ww, Co, k, s, p = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000)
    x = _bf(torch.randn(B, Ci, Hh, Ww, generator=g))
    w = _bf(torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5)
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x, w, b, s, p)
    y = _run_conv(x, w, b, s, p, out_f32=True)
    # f32 accumulate of exactly representable products: error is summation order only
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=2e-4)
    yb = _run_conv(x, w, b, s, p, relu=True)
    torch.testing.assert_close(yb, F.relu(ref), rtol=2 ** -7, atol=2e-3)   # + one bf16 rounding


@pytest.mark.parametrize("variant", [0x10000000, 0x08000000, 0x04000000, 0x14000000, 0x0c000000,   # K32 / K64 / BIG_TILES and pairs
                                     0x01000000, 0x11000000, 0x09000000,    # these three: LDS-staged epilogue
                                     0x0c400000,                            # 256x256 8-wave tiles (forced)
                                     0x0c440000, 0x08040000])               # hand-placed K step: 256x256 tiles / 128x128 tiles
def test_conv_loader_variants(variant):
    """The launch-plan selectors of include/sipmask_hip.h (forced 32- / 64-wide K steps, big tiles, LDS-staged epilogue,
    256x256 tiles, hand-placed K step) must give the same results as the default plan on every tile shape, incl. residual
    and ragged tiles.  (The rejected variants of rounds 1-2 are not in the default build: csrc/experiments.h.)"""
    g = torch.Generator().manual_seed(17)
    for (B, Ci, Hh, Ww, Co, k, s, p) in [(2, 64, 17, 23, 128, 3, 1, 1), (2, 128, 20, 28, 64, 1, 1, 0),
                                          (2, 64, 21, 19, 5, 3, 1, 1), (2, 3, 37, 45, 64, 7, 2, 3),
                                          (1, 768, 10, 14, 512, 1, 1, 0), (1, 256, 25, 42, 256, 3, 2, 1)]:
        x = _bf(torch.randn(B, Ci, Hh, Ww, generator=g))
        w = _bf(torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5)
        b = torch.randn(Co, generator=g)
        ref = F.conv2d(x, w, b, s, p)
        r = _bf(torch.randn_like(ref))
        y = _run_conv(x, w, b, s, p, relu=True, out_f32=True, residual=r, extra_flags=variant)
        torch.testing.assert_close(y, F.relu(ref + r), rtol=1e-4, atol=2e-4)
        yb = _run_conv(x, w, b, s, p, extra_flags=variant)
        torch.testing.assert_close(yb, ref, rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("ring", [3, 4])
def test_conv_k32_operand_ring_is_bit_identical(ring, monkeypatch):
    """EXPERIMENTS build only (`make -C sipmask_amd/csrc EXPERIMENTS=1`; skipped on the default library): the 32-wide-K kernel
    with a 3- / 4-stage LDS ring (counted vmcnt waits + raw barrier per K step, round 5; SIPMASK_EXP_K32_RING) accumulates every
    output in the same K order as the two-stage loop: identical bits on all three tiles it takes (128x128 via BIG_TILES,
    128x64 / 64x64 by the occupancy rule), with the prefetched same-row residual, ragged position tiles, a 3x3 (the flat
    loader's tap walk) and bf16 / f32 outputs; a launch it does not take (fewer than 8 K steps) keeps the two-stage loop."""
    g = torch.Generator().manual_seed(50 + ring)
    BIG = 0x04000000
    seen = set()
    for (B, Ci, Hh, Ww, Co, k, s, p, fl) in [(2, 1024, 21, 33, 256, 1, 1, 0, 0), (2, 1024, 21, 33, 256, 1, 1, 0, BIG),
                                              (2, 256, 50, 84, 1024, 1, 1, 0, 0), (2, 256, 19, 23, 128, 1, 1, 0, BIG),
                                              (2, 512, 25, 42, 64, 1, 1, 0, 0), (1, 64, 30, 41, 128, 3, 1, 1, BIG),
                                              (1, 64, 12, 10, 256, 1, 1, 0, 0)]:
        x = _bf(torch.randn(B, Ci, Hh, Ww, generator=g))
        w = _bf(torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5)
        b = torch.randn(Co, generator=g)
        r = _bf(torch.randn(B, Co, Hh, Ww, generator=g))
        for out_f32, res in ((True, r), (False, None), (False, r)):
            pl = {}
            monkeypatch.setenv("SIPMASK_EXP_K32_RING", "0")
            y0 = _run_conv(x, w, b, s, p, relu=True, out_f32=out_f32, residual=res, extra_flags=fl)
            monkeypatch.setenv("SIPMASK_EXP_K32_RING", str(ring))
            y1 = _run_conv(x, w, b, s, p, relu=True, out_f32=out_f32, residual=res, extra_flags=fl, want_plan=pl)
            if pl["ring_stages"] == 0 and Ci * k * k >= 256:
                pytest.skip("default build: the operand ring exists in the experiments build only")
            assert pl["k_step"] == 32
            assert pl["ring_stages"] == (ring if Ci * k * k >= 256 else 0), pl
            if pl["ring_stages"]:
                seen.add((pl["tile_cout"], pl["tile_pos"]))
            assert torch.equal(y0, y1), (Ci, Co, k, fl, out_f32, float((y0 - y1).abs().max()))
        ref = F.relu(F.conv2d(x, w, b, s, p) + r)
        torch.testing.assert_close(_run_conv(x, w, b, s, p, relu=True, out_f32=True, residual=r, extra_flags=fl), ref,
                                   rtol=1e-4, atol=2e-4)
    assert seen == {(128, 128), (128, 64), (64, 64)}, seen


@pytest.mark.parametrize("cfg", [
    (2, 512, 25, 42, 512, 3, 1, 1),     # layer4 conv2 at B=2: 3x3, K = 4608 (72 K steps), M = 2100
    (4, 512, 25, 42, 512, 3, 1, 1),     # the same at B=4 (132 tiles of 128x128 x 4 slices)
    (1, 256, 50, 84, 256, 3, 1, 1),     # layer3 conv2 at B=1 (K = 2304: 36 K steps)
    (1, 256, 13, 21, 256, 3, 1, 1),     # FPN output conv on a tiny map
])
def test_conv_split_k_vs_torch(cfg):
    """sm_conv2d_ws: under-filled launches with long K loops run as S K-slices per (larger) tile + a reduce/epilogue
    kernel; same tolerance as the unsplit kernel (f32 accumulation, different summation order), with bias, same-row
    residual and ReLU in the reduce kernel, bf16 and f32 outputs; without a workspace the call is sm_conv2d."""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    B, Ci, Hh, Ww, Co, k, st, p = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = _bf(torch.randn(B, Ci, Hh, Ww, generator=g))
    w = _bf(torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5)
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x, w, b, st, p)
    r = _bf(torch.randn_like(ref))
    xh = torch.empty(B * Hh * Ww, Ci, dtype=torch.bfloat16, device=dev)
    H.nchw_to_nhwc_bf16(x.to(dev).contiguous(), xh, Ci)
    wq, co_pad = H.prep_conv_weight(w.to(dev), Ci)
    res = r.permute(0, 2, 3, 1).reshape(-1, Co).to(torch.bfloat16).to(dev).contiguous()
    for out_f32 in (True, False):
        flags = _lib.SM_CONV_RELU | _lib.SM_CONV_RES_ADD | (_lib.SM_CONV_OUT_F32 if out_f32 else 0)
        d = H.make_conv_desc(B, [(Hh, Ww)], [(Hh, Ww)], [0], [0], Ci, Co, co_pad, k, st, p, Ci, Co, flags=flags, res_cstride=Co)
        pl = H.conv_plan(d)
        assert pl["split_k"] > 1 and pl["workspace_bytes"] > 0, pl
        ws = torch.empty(pl["workspace_bytes"], dtype=torch.uint8, device=dev)
        y = torch.full((B * Hh * Ww, Co), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
        H.conv2d_ws(d, xh, wq, b.to(dev), res, y, ws)
        torch.cuda.synchronize()
        got = y.float().view(B, Hh, Ww, Co).permute(0, 3, 1, 2).cpu()
        if out_f32:
            torch.testing.assert_close(got, F.relu(ref + r), rtol=1e-4, atol=2e-4)
        else:
            torch.testing.assert_close(got, F.relu(ref + r), rtol=2 ** -7, atol=2e-3)
        y2 = torch.empty_like(y)
        H.conv2d_ws(d, xh, wq, b.to(dev), res, y2, None)                 # no workspace: the unsplit plan
        torch.cuda.synchronize()
        torch.testing.assert_close(y2.float(), y.float(), rtol=2 ** -7 if not out_f32 else 1e-4, atol=2e-3 if not out_f32 else 2e-4)


def test_conv_residual_and_input_relu():
    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn(2, 64, 12, 10, generator=g))
    w = _bf(torch.randn(256, 64, 1, 1, generator=g) / 8)
    r = _bf(torch.randn(2, 256, 12, 10, generator=g))
    y = _run_conv(x, w, None, 1, 0, relu=True, out_f32=True, residual=r)
    torch.testing.assert_close(y, F.relu(F.conv2d(x, w) + r), rtol=1e-4, atol=2e-4)
    w3 = _bf(torch.randn(64, 64, 3, 3, generator=g) / 24)
    y = _run_conv(x, w3, None, 2, 1, out_f32=True, in_relu=True)
    torch.testing.assert_close(y, F.conv2d(F.relu(x), w3, None, 2, 1), rtol=1e-4, atol=2e-4)


def test_conv_multilevel_scale_and_nearest_residual():
    """One launch over 3 levels with per-level Scale on the first 4 channels; and the FPN
    top-down nearest-neighbour residual."""
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    B, C = 2, 64
    sizes = [(12, 20), (6, 10), (3, 5)]
    lv = H.Levels(B, sizes)
    xs = [_bf(torch.randn(B, C, h, w, generator=g)) for h, w in sizes]
    w = _bf(torch.randn(5, C, 3, 3, generator=g) / 24)
    bias = torch.randn(5, generator=g)
    scales = [1.5, 0.5, 2.0]
    x = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, C) for t in xs]).to(torch.bfloat16).to(dev)
    wq, co_pad = H.prep_conv_weight(w.to(dev))
    y = torch.zeros(lv.rows, 8, dtype=torch.float32, device=dev)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, C, 5, co_pad, 3, 1, 1, C, 8, flags=_lib.SM_CONV_OUT_F32,
                         scale_nch=4, level_scale=scales)
    H.conv2d(d, x, wq, bias.to(dev), None, y)
    torch.cuda.synchronize()
    for l, (h, wd) in enumerate(sizes):
        ref = F.conv2d(xs[l], w, bias, 1, 1)
        ref[:, :4] *= scales[l]
        got = y[lv.row0[l]:lv.row0[l] + B * h * wd].view(B, h, wd, 8).permute(0, 3, 1, 2)[:, :5].cpu()
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-4)
    # nearest residual: lateral(C4) + up(lateral(C5)), fpn.py:149-152
    coarse = _bf(torch.randn(B, 128, 7, 9, generator=g))
    fine = _bf(torch.randn(B, C, 13, 18, generator=g))
    wl = _bf(torch.randn(128, C, 1, 1, generator=g) / 8)
    ref = F.conv2d(fine, wl) + F.interpolate(coarse, size=(13, 18), mode="nearest")
    xf = fine.permute(0, 2, 3, 1).reshape(-1, C).to(torch.bfloat16).to(dev)
    rc = coarse.permute(0, 2, 3, 1).reshape(-1, 128).to(torch.bfloat16).to(dev)
    wq, co_pad = H.prep_conv_weight(wl.to(dev))
    y = torch.zeros(B * 13 * 18, 128, dtype=torch.float32, device=dev)
    d = H.make_conv_desc(B, [(13, 18)], [(13, 18)], [0], [0], C, 128, co_pad, 1, 1, 0, C, 128,
                         flags=_lib.SM_CONV_OUT_F32 | _lib.SM_CONV_RES_NEAREST, res_cstride=128, res_sizes=[(7, 9)],
                         res_row0=[0])
    H.conv2d(d, xf, wq, None, rc, y)
    torch.cuda.synchronize()
    torch.testing.assert_close(y.view(B, 13, 18, 128).permute(0, 3, 1, 2).cpu(), ref, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("extra", [0, 0x04000000, 0x01000000, 0x04400000])   # default / big tiles / LDS-staged epilogue / 256x256 (forced)
def test_conv_fused_groupnorm_statistics(extra):
    """sm_conv2d_gn_stats: per (image, level, group of 8 channels) sum / sum of squares of the conv
    output accumulated in the epilogue, incl. tiles that span several images (tiny levels)."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(8)
    B, C, Co = 3, 64, 256
    sizes = [(21, 30), (9, 11), (3, 5), (2, 2)]
    lv = H.Levels(B, sizes)
    xs = [_bf(torch.randn(B, C, h, w, generator=g)) for h, w in sizes]
    w = _bf(torch.randn(Co, C, 3, 3, generator=g) / 24)
    x = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, C) for t in xs]).to(torch.bfloat16).to(dev)
    wq, co_pad = H.prep_conv_weight(w.to(dev))
    y = torch.zeros(lv.rows, Co, dtype=torch.bfloat16, device=dev)
    stats = torch.full((B, len(sizes), Co // 8, 2), 123, dtype=torch.int64, device=dev)    # zeroed by the call
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, C, Co, co_pad, 3, 1, 1, C, Co, flags=extra)
    H.conv2d_gn_stats(d, x, None, wq, None, None, y, stats)
    torch.cuda.synchronize()
    for l, (h, wd) in enumerate(sizes):
        ref = F.conv2d(xs[l], w, None, 1, 1).double()                       # [B,Co,h,w]
        rs = ref.view(B, Co // 8, 8 * h * wd)
        got = H.gn_stats_to_float(stats[:, l].cpu())
        torch.testing.assert_close(got[..., 0], rs.sum(-1), rtol=1e-4, atol=2e-3)
        torch.testing.assert_close(got[..., 1], (rs * rs).sum(-1), rtol=1e-4, atol=2e-3)
        out = y[lv.row0[l]:lv.row0[l] + B * h * wd].float().view(B, h, wd, Co).permute(0, 3, 1, 2).cpu()
        torch.testing.assert_close(out, ref.float(), rtol=2 ** -7, atol=2e-3)
    # normalisation with those statistics == GroupNorm(32) + ReLU
    gamma, beta = torch.randn(Co, generator=g), torch.randn(Co, generator=g)
    H.groupnorm_apply(y, y, gamma.to(dev), beta.to(dev), stats.view(-1), lv, Co, 32, 1e-5, True)
    torch.cuda.synchronize()
    for l, (h, wd) in enumerate(sizes):
        ref = F.relu(F.group_norm(F.conv2d(xs[l], w, None, 1, 1), 32, gamma, beta, 1e-5))
        out = y[lv.row0[l]:lv.row0[l] + B * h * wd].float().view(B, h, wd, Co).permute(0, 3, 1, 2).cpu()
        torch.testing.assert_close(out, ref, rtol=2e-2, atol=3e-2)


@pytest.mark.parametrize("extra,shared_x", [(0, True), (0, False), (0x04400000, True), (0x04400000, False), (0x04440000, False)])
def test_grouped_conv_with_groupnorm_statistics(extra, shared_x):
    """group dimension of sm_conv_desc: two convs of identical shape (own weights, own outputs and GN statistics,
    shared or own inputs) in ONE launch == the two launches, on the default tiles and on the 256x256 8-wave tile"""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    B, C, Co, G = 2, 64, 256, 2
    sizes = [(19, 27), (9, 11), (3, 5)]
    lv = H.Levels(B, sizes)
    xs = [[_bf(torch.randn(B, C, h, w, generator=g)) for h, w in sizes] for _ in range(1 if shared_x else G)]
    ws = [_bf(torch.randn(Co, C, 3, 3, generator=g) / 24) for _ in range(G)]
    flat = lambda t: torch.cat([a.permute(0, 2, 3, 1).reshape(-1, C) for a in t])
    x = torch.cat([flat(t) for t in xs]).to(torch.bfloat16).to(dev)
    packed = [H.prep_conv_weight(w.to(dev)) for w in ws]
    wq, co_pad = torch.stack([p[0] for p in packed]).contiguous(), packed[0][1]
    y = torch.zeros(G * lv.rows, Co, dtype=torch.bfloat16, device=dev)
    S = 2 * B * len(sizes) * (Co // 8)
    stats = torch.full((G * S,), 7, dtype=torch.int64, device=dev)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, C, Co, co_pad, 3, 1, 1, C, Co, flags=extra, ngroups=G,
                         x_group_rows=0 if shared_x else lv.rows, y_group_rows=lv.rows, w_group_stride=packed[0][0].numel(),
                         gn_group_stride=S)
    H.conv2d_gn_stats(d, x, None, wq, None, None, y, stats)
    torch.cuda.synchronize()
    for gi in range(G):
        src = xs[0] if shared_x else xs[gi]
        st = H.gn_stats_to_float(stats[gi * S:(gi + 1) * S].view(B, len(sizes), Co // 8, 2).cpu())
        for l, (h, wd) in enumerate(sizes):
            ref = F.conv2d(src[l], ws[gi], None, 1, 1).double()
            r0 = gi * lv.rows + lv.row0[l]
            out = y[r0:r0 + B * h * wd].float().view(B, h, wd, Co).permute(0, 3, 1, 2).cpu()
            torch.testing.assert_close(out, ref.float(), rtol=2 ** -7, atol=2e-3)
            rs = ref.view(B, Co // 8, 8 * h * wd)
            torch.testing.assert_close(st[:, l, :, 0], rs.sum(-1), rtol=1e-4, atol=2e-3)
            torch.testing.assert_close(st[:, l, :, 1], (rs * rs).sum(-1), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("shape", [(2, 256, 14, 19, 256), (1, 64, 9, 9, 40)])
def test_deform_conv_vs_oracle(shape):
    from sipmask_amd import hip_ops as H, _lib
    dev = _dev()
    B, C, Hh, Ww, Co = shape
    g = torch.Generator().manual_seed(11)
    G = 4
    x = _bf(torch.randn(B, C, Hh, Ww, generator=g))
    w = _bf(torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    off = torch.randn(B, G * 18, Hh, Ww, generator=g) * 1.5
    off[0, :, 0, 0] = 0.0            # exact-integer sample
    off[0, 0::2, 1, 1] = -2.0        # h_im == -1 for the top taps -> must be excluded (> -1)
    ref = O.deform_conv(x, off, w, 1, 1, 1, G)
    xh = torch.empty(B * Hh * Ww, C, dtype=torch.bfloat16, device=dev)
    H.nchw_to_nhwc_bf16(x.to(dev), xh, C)
    offr = off.permute(0, 2, 3, 1).reshape(-1, G * 18).contiguous().to(dev)
    wq, co_pad = H.prep_conv_weight(w.to(dev))
    y = torch.empty(B * Hh * Ww, Co, dtype=torch.float32, device=dev)
    d = H.make_conv_desc(B, [(Hh, Ww)], [(Hh, Ww)], [0], [0], C, Co, co_pad, 3, 1, 1, C, Co,
                         flags=_lib.SM_CONV_OUT_F32, deform_groups=G)
    H.deform_conv2d(d, xh, offr, wq, None, y)
    torch.cuda.synchronize()
    got = y.view(B, Hh, Ww, Co).permute(0, 3, 1, 2).cpu()
    # (1) the arithmetic of the bf16 plan, exactly: corners blended in f32, the SAMPLE rounded once to bf16 (the MFMA
    # operand type), f32 accumulation -- against the oracle with the same single rounding of its sampled columns.  What is
    # left is accumulation order and the rare sample whose f32 blend lands on the other side of a bf16 rounding boundary
    # because the two sides associate the four products differently (measured max ~5e-4).
    ref_bf = O.deform_conv(x, off, w, 1, 1, 1, G, col_round=lambda t: t.to(torch.bfloat16).float())
    torch.testing.assert_close(got, ref_bf, rtol=2e-3, atol=2e-3)
    # (2) against the plain f32 oracle the operand rounding shows: rel 2^-9 per sample, averaged over K = 9*C products
    # (the f32 plan, tests/test_gpu_f32_plan.py, has no such term)
    torch.testing.assert_close(got, ref, rtol=2e-2, atol=1.5e-2)
    assert float((got - ref).abs().mean()) < 2e-3
    # zero offsets: the gather is exact, so the result must match the plain conv tightly
    H.deform_conv2d(d, xh, torch.zeros_like(offr), wq, None, y)
    torch.cuda.synchronize()
    torch.testing.assert_close(y.view(B, Hh, Ww, Co).permute(0, 3, 1, 2).cpu(), F.conv2d(x, w, None, 1, 1),
                               rtol=1e-4, atol=2e-4)


def test_groupnorm_relu_multilevel():
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    B, C = 2, 256
    sizes = [(20, 33), (10, 17), (5, 9), (3, 5), (2, 3)]
    lv = H.Levels(B, sizes)
    xs = [_bf(torch.randn(B, C, h, w, generator=g) * 3 + 0.5) for h, w in sizes]
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    x = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, C) for t in xs]).to(torch.bfloat16).to(dev)
    stats = H.gn_stats_alloc(B * 5 * 32, dev)
    H.groupnorm(x, x, gamma.to(dev), beta.to(dev), stats, lv, C, 32, 1e-5, True)
    torch.cuda.synchronize()
    for l, (h, w) in enumerate(sizes):
        ref = F.relu(F.group_norm(xs[l], 32, gamma, beta, 1e-5))
        got = x[lv.row0[l]:lv.row0[l] + B * h * w].float().view(B, h, w, C).permute(0, 3, 1, 2).cpu()
        torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=1e-2)


def test_maxpool_upsample_layout():
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    x = _bf(torch.randn(2, 64, 21, 30, generator=g))
    xh = torch.empty(2 * 21 * 30, 64, dtype=torch.bfloat16, device=dev)
    H.nchw_to_nhwc_bf16(x.to(dev), xh, 64)
    torch.testing.assert_close(xh.float().view(2, 21, 30, 64).permute(0, 3, 1, 2).cpu(), x, rtol=0, atol=0)
    y = torch.empty(2 * 11 * 15, 64, dtype=torch.bfloat16, device=dev)
    H.maxpool3x3s2(xh, y, 2, 21, 30, 64)
    torch.testing.assert_close(y.float().view(2, 11, 15, 64).permute(0, 3, 1, 2).cpu(), F.max_pool2d(x, 3, 2, 1),
                               rtol=0, atol=0)
    for f in (1, 2, 4):
        out = torch.zeros(2 * 21 * f * 30 * f, 128, dtype=torch.bfloat16, device=dev)
        H.upsample_bilinear(xh, out, 2, 21, 30, 64, f, 64, 128, 64, False)
        ref = x if f == 1 else F.interpolate(x, scale_factor=f, mode="bilinear", align_corners=False)
        got = out.float().view(2, 21 * f, 30 * f, 128).permute(0, 3, 1, 2).cpu()
        torch.testing.assert_close(got[:, 64:], ref, rtol=2 ** -7, atol=1e-2)
        assert float(got[:, :64].abs().max()) == 0
    xf = torch.randn(1, 32, 9, 13, generator=g)
    src = xf.permute(0, 2, 3, 1).reshape(-1, 32).contiguous().to(dev)
    out = torch.empty(36 * 52, 32, device=dev)
    H.upsample_bilinear(src, out, 1, 9, 13, 32, 4, 32, 32, 0, True)
    torch.testing.assert_close(out.view(1, 36, 52, 32).permute(0, 3, 1, 2).cpu(),
                               F.interpolate(xf, scale_factor=4, mode="bilinear", align_corners=False),
                               rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 75, 131), (1, 224, 320), (3, 33, 57), (1, 7, 7)])
def test_stem_fused_vs_torch(shape):
    """sm_stem_fused: conv1 7x7/2 + folded bn1 + ReLU + maxpool 3x3/2 (resnet.py:497-505) in one launch, against torch
    f32 on the same bf16-rounded operands.  Sizes that leave partial 8 x 14 pooled tiles on both axes, an image smaller
    than one tile, several images (tile -> image decoding).  Bound: one bf16 ulp (the f32 sums differ in order only)."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    B, Hi, Wi = shape
    g = torch.Generator().manual_seed(Hi * 1000 + Wi)
    img = torch.randn(B, 3, Hi, Wi, generator=g) * 1.5
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.08
    b = torch.randn(64, generator=g) * 0.3
    h1, w1 = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
    h2, w2 = (h1 - 1) // 2 + 1, (w1 - 1) // 2 + 1
    y = torch.full((B * h2 * w2, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    wp = H.prep_stem_weight(w.to(dev))
    assert tuple(wp.shape) == (64, 7, 8, 4) and float(wp[:, :, 7].abs().max()) == 0 and float(wp[..., 3].abs().max()) == 0
    H.stem_fused(img.to(dev), wp, b.to(dev), y)
    ref = F.max_pool2d(_bf(F.relu(F.conv2d(_bf(img).double(), _bf(w).double(), b.double(), 2, 3)).float()), 3, 2, 1)
    got = y.float().view(B, h2, w2, 64).permute(0, 3, 1, 2).cpu()
    assert tuple(ref.shape) == tuple(got.shape)
    assert bool(torch.isfinite(got).all())
    diff = (got - ref).abs()
    assert bool((diff <= 2.0 ** -7 * ref.abs() + 1e-30).all()), float(diff.max())
    assert float((diff > 0).float().mean()) < 0.02          # rounding-boundary cases only


# ------------------------------------------------------------------------------------- NMS
def _kat(golden_dir):
    return json.load(open(os.path.join(golden_dir, "nms_kat.json")))


def test_nms_golden_vectors(golden_dir):
    """The reference's own keep-index vectors (B/tests/test_nms.py) through sm_nms: bit-exact."""
    from sipmask_amd import ops as P
    dev = _dev()
    kat = _kat(golden_dir)
    c = kat["caffe2_5box"]
    dets = torch.tensor(c["dets"], dtype=torch.float32, device=dev)
    for thr, gt in zip(c["thresh"], c["keep"]):
        kept, inds = P.nms(dets, thr)
        assert inds.cpu().tolist() == gt
        assert torch.equal(kept, dets[inds])
    c = kat["boxes53"]
    dets = torch.cat([torch.tensor(c["boxes"]), torch.tensor(c["scores"])[:, None]], 1).float().to(dev)
    assert P.nms(dets, c["thresh"])[1].cpu().tolist() == c["keep"]
    for name in ("wrapper_doctest", "mmdet_test4"):
        c = kat[name]
        assert len(P.nms(torch.tensor(c["dets"], dtype=torch.float32, device=dev), c["thresh"])[1]) == c["n_keep"]
    e = torch.zeros(0, 5, device=dev)
    assert P.nms(e, 0.5)[1].numel() == 0
    # exact-threshold pair: GPU rule is '>' (nms_kernel.cu:61) so both survive
    d2 = torch.tensor([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8]], dtype=torch.float32, device=dev)
    assert P.nms(d2, 0.5)[1].cpu().tolist() == [0, 1]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 700, 3000])
def test_nms_random_vs_oracle(n):
    from sipmask_amd import ops as P
    dev = _dev()
    rng = np.random.RandomState(n)
    xy = rng.rand(n, 2).astype(np.float32) * 300
    wh = rng.rand(n, 2).astype(np.float32) * 80 + 2
    sc = rng.rand(n).astype(np.float32)
    sc[: n // 3] = np.round(sc[: n // 3], 1)          # many exact score ties
    dets = np.concatenate([xy, xy + wh, sc[:, None]], 1).astype(np.float32)
    ref = O.nms(dets, 0.5, "gpu")
    got = P.nms(torch.from_numpy(dets).to(dev), 0.5)[1].cpu().numpy()
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("n", [64, 700])
def test_nms_inverted_boxes_vs_oracle(n):
    """SipMask's bbox_pred is neither exp'd nor ReLU'd, so decoded boxes with x2 < x1 - 1 (negative "+1" area)
    are reachable.  devIoU (nms_kernel.cu:14-22) then divides by a union that may be negative or zero and the quotient
    compares as it falls (negative / inf / NaN never exceed thr ... or do): the division-free fast path must not decide
    those pairs.  A third of the boxes are inverted along x, y or both, some with a union of exactly 0."""
    from sipmask_amd import ops as P
    dev = _dev()
    rng = np.random.RandomState(1000 + n)
    xy = rng.rand(n, 2).astype(np.float32) * 200
    wh = rng.rand(n, 2).astype(np.float32) * 80 + 2
    inv = rng.rand(n, 2) < 0.2
    wh = np.where(inv, -wh * 3 - 2, wh).astype(np.float32)
    dets = np.concatenate([xy, xy + wh, rng.rand(n, 1).astype(np.float32)], 1).astype(np.float32)
    dets[3, :4] = [10, 10, 9, 30]            # width + 1 == 0: area 0
    dets[4, :4] = [50, 50, 20, 60]           # strongly negative area: union with most boxes < 0
    dets[3:5, 4] = [0.999, 0.998]            # ... and kept early, so every later box is tested against them
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = O.nms(dets, 0.5, "gpu")
    got = P.nms(torch.from_numpy(dets).to(dev), 0.5)[1].cpu().numpy()
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("case", [(500, 80, 100, 0.05), (3350, 80, 100, 0.05), (300, 7, 20, 0.3), (200, 3, 100, 0.99),
                                  (900, 80, 2000, 0.02)])
def test_multiclass_nms_idx_vs_oracle(case):
    from sipmask_amd import ops as P
    dev = _dev()
    K, C, max_num, thr = case
    rng = np.random.RandomState(K + C)
    xy = rng.rand(K, 2).astype(np.float32) * 600
    wh = rng.rand(K, 2).astype(np.float32) * 120 + 4
    boxes = np.concatenate([xy, xy + wh], 1)
    scores = (rng.rand(K, C + 1).astype(np.float32) ** 6)
    scores[:, 0] = 0
    scores[: K // 4] = np.round(scores[: K // 4], 2)    # ties
    ctr = rng.rand(K).astype(np.float32)
    rb, rl, rk = O.multiclass_nms_idx(boxes, scores, thr, 0.5, max_num, ctr)
    b, l, k = P.multiclass_nms_idx(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), thr,
                                   dict(type="nms", iou_thr=0.5), max_num, torch.from_numpy(ctr).to(dev))
    np.testing.assert_array_equal(k.cpu().numpy(), rk)       # bit-exact keep indices
    np.testing.assert_array_equal(l.cpu().numpy(), rl)
    np.testing.assert_array_equal(b.cpu().numpy(), rb)       # boxes copied, score = f32 product: exact


@pytest.mark.parametrize("case", [(3350, 3, 600.0, (513, 576, 3350)), (1500, 4, 150.0, (577, 640, 1, 1500)),
                                  (5000, 2, 400.0, (5000, 4097)), (6000, 40, 900.0, (700,) * 40)])
def test_multiclass_nms_heavy_classes_vs_oracle(case):
    """Classes with more than 512 candidates above score_thr leave the class block: sort there, IoU bit matrix over the
    whole chip (nms_heavy_matrix_kernel), one-wave scan (nms_heavy_scan_kernel) -- nms_kernel.cu:24-68,113-138.  Class
    sizes on and around the 64-row chunk edges, every candidate in one class, crowded boxes (long suppression chains),
    duplicates and score ties, kmax > 4096 (two mask words per lane), and more heavy classes than slots (40 > 32: the
    rest stay in the block)."""
    from sipmask_amd import ops as P
    dev = _dev()
    K, C, extent, sizes = case
    rng = np.random.RandomState(K + C)
    xy = rng.rand(K, 2).astype(np.float32) * extent
    wh = rng.rand(K, 2).astype(np.float32) * 120 + 4
    boxes = np.concatenate([xy, xy + wh], 1)
    boxes[K // 2: K // 2 + 40] = boxes[:40]                    # duplicates
    scores = np.zeros((K, C + 1), np.float32)
    for c, n in enumerate(sizes):
        sel = rng.permutation(K)[:n]
        scores[sel, c + 1] = 0.06 + 0.9 * rng.rand(n).astype(np.float32)
    scores[: K // 4] = np.round(scores[: K // 4], 2)           # ties
    ctr = 0.5 + 0.5 * rng.rand(K).astype(np.float32)
    for max_num in (100, 2000):
        rb, rl, rk = O.multiclass_nms_idx(boxes, scores, 0.05, 0.5, max_num, ctr)
        b, l, k = P.multiclass_nms_idx(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 0.05,
                                       dict(type="nms", iou_thr=0.5), max_num, torch.from_numpy(ctr).to(dev))
        np.testing.assert_array_equal(k.cpu().numpy(), rk)
        np.testing.assert_array_equal(l.cpu().numpy(), rl)
        np.testing.assert_array_equal(b.cpu().numpy(), rb)


@pytest.mark.parametrize("case", [(700, 80, 0.1), (150, 5, 0.0), (1, 3, 0.1), (3000, 20, 0.3)])
def test_fast_nms_vs_oracle(case):
    """SipMaskHead.fast_nms (ssd_flag configs, sipmask_head.py:868-910): same boxes, classes, coefficient rows."""
    from sipmask_amd import ops as P
    dev = _dev()
    K, C, thr = case
    rng = np.random.RandomState(K * 3 + C)
    xy = rng.rand(K, 2).astype(np.float32) * 300
    wh = rng.rand(K, 2).astype(np.float32) * 120 + 4
    boxes = np.concatenate([xy, xy + wh], 1)
    if K > 10:
        boxes[5] = boxes[4]                               # exact duplicates
        boxes[7, 2:] = boxes[7, :2]                       # two degenerate (zero-area) boxes: their mutual IoU is
        boxes[9, 2:] = boxes[9, :2]                       # 0/0 = NaN, which torch.max propagates (-> not kept)
    scores = (rng.rand(C, K).astype(np.float32) ** 3)
    scores[:, : K // 4] = np.round(scores[:, : K // 4], 2)   # ties
    cofs = rng.randn(K, 128).astype(np.float32)
    rb, rl, rm = O.fast_nms(boxes, scores, cofs, 0.5, 200, thr, 100)
    b, l, m = P.fast_nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev),
                         torch.from_numpy(cofs).to(dev), 0.5, 200, thr)
    np.testing.assert_array_equal(l.cpu().numpy(), rl)
    np.testing.assert_array_equal(b.cpu().numpy(), rb)
    np.testing.assert_array_equal(m.cpu().numpy(), rm)


@pytest.mark.parametrize("case", [
    # (ho, wo, canvas_h, canvas_w, kind)
    (64, 96, 64, 96, "blobs"), (64, 96, 60, 91, "blobs"), (50, 75, 64, 96, "blobs"), (37, 53, 37, 53, "noise"),
    (200, 336, 200, 333, "boxes"), (800, 1344, 800, 1333, "boxes"), (16, 16, 16, 16, "edge")])
def test_rle_encode_vs_oracle(case):
    """Device COCO RLE (sm_rle_encode) == the oracle's restatement of maskApi.c on pasted masks: run lengths and
    compressed strings, with and without the box hint, aligned and unaligned rows, canvas smaller/larger.
    PARITY UNPINNED: neither pycocotools nor any RLE string exists in this image or under the reference
    tree, so the oracle side is held by brute-force run lengths and encode -> string -> parse -> decode round trips only
    (tests/test_oracle_ops.py: test_rle_restatement_round_trip); this test proves HIP == restatement, not == pycocotools."""
    from sipmask_amd import hip_ops as H, ops as P
    dev = _dev()
    ho, wo, ch, cw, kind = case
    rng = np.random.RandomState(ho * 7 + wo)
    B, n = 2, 6
    m = np.zeros((B, n, ho, wo), np.uint8)
    rect = np.zeros((B, n, 4), np.int32)
    for b in range(B):
        for i in range(n):
            x0, x1 = sorted(rng.randint(0, wo + 1, 2))
            y0, y1 = sorted(rng.randint(0, ho + 1, 2))
            if kind == "edge":      # boxes touching the borders, single rows/columns, empty and full masks
                x0, y0, x1, y1 = [(0, 0, wo, ho), (0, 0, 1, ho), (wo - 1, 0, wo, ho), (0, ho - 1, wo, ho),
                                  (3, 3, 3, 3), (5, 0, 9, ho)][i]
            if kind == "noise":
                m[b, i] = rng.rand(ho, wo) < 0.5
                x0, y0, x1, y1 = 0, 0, wo, ho
            elif kind == "blobs":
                yy, xx = np.mgrid[:ho, :wo]
                blob = ((yy - (y0 + y1) / 2) ** 2 * 1.3 + (xx - (x0 + x1) / 2) ** 2) < (max(x1 - x0, 2) / 2) ** 2
                m[b, i, y0:y1, x0:x1] = (blob & (rng.rand(ho, wo) < 0.97))[y0:y1, x0:x1]
            else:
                m[b, i, y0:y1, x0:x1] = 1
                if i % 2:
                    m[b, i, y0:y1, x0:x1] &= (rng.rand(y1 - y0, x1 - x0) < 0.9)
            rect[b, i] = [x0 - rng.randint(0, 3), y0 - rng.randint(0, 3), x1 + rng.randint(0, 3), y1 + rng.randint(0, 3)]
    ndet = np.array([n, n - 2], np.int32)
    mt, nt, rt = torch.from_numpy(m).to(dev), torch.from_numpy(ndet).to(dev), torch.from_numpy(rect).to(dev)
    def pasted(x):
        im = np.zeros((ch, cw), np.uint8)
        im[:min(ho, ch), :min(wo, cw)] = x[:ch, :cw]
        return im
    need = max(len(O.rle_counts(pasted(m[b, i]))) for b in range(B) for i in range(n))
    for hint in (None, rt):
        out = H.rle_alloc(B, n, cw, dev, max_runs=need + 1)
        H.rle_encode(mt, nt, (ch, cw), out, hint)
        got = H.rle_fetch(out, B, n, ndet, (ch, cw))
        nruns = out["nruns"].cpu().numpy().reshape(B, n)
        counts = out["counts"].cpu().numpy().view(np.uint32).reshape(B, n, -1)
        for b in range(B):
            assert len(got[b]) == ndet[b]
            for i in range(n):
                if i >= ndet[b]:
                    assert nruns[b, i] == 0
                    continue
                ref = O.paste_and_encode(m[b, i], (ch, cw))
                im = np.zeros((ch, cw), np.uint8)
                im[:min(ho, ch), :min(wo, cw)] = m[b, i][:ch, :cw]
                rc = O.rle_counts(im)
                assert nruns[b, i] == len(rc)
                np.testing.assert_array_equal(counts[b, i, :len(rc)], np.array(rc, np.uint32))
                assert got[b][i]["counts"] == ref["counts"], (b, i)
                assert got[b][i]["size"] == [ch, cw]
    # the convenience wrapper grows max_runs when a mask needs more
    res = P.encode_masks(mt, nt, (ch, cw), max_runs=8)
    for b in range(B):
        for i in range(ndet[b]):
            assert res[b][i]["counts"] == O.paste_and_encode(m[b, i], (ch, cw))["counts"]


def test_mask_rects_cover_assembled_masks():
    """sm_mask_rects is a safe hint: every non-zero pixel sm_mask_assemble writes lies inside the rectangle."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, K, n, hm, wm = 1, 40, 40, 64, 96
    basis = torch.randn(B, 32, hm, wm, generator=g).to(dev)
    cofs = (torch.randn(B, K, 128, generator=g) * 0.5).to(dev)
    xy = torch.rand(B, n, 2, generator=g) * torch.tensor([2.0 * wm, 2.0 * hm]) - 10
    wh = torch.rand(B, n, 2, generator=g) * 60 + 1
    det = torch.cat([xy, xy + wh, torch.rand(B, n, 1, generator=g)], 2).to(dev)
    keep = torch.arange(n).view(1, n).to(dev)
    ndet = torch.tensor([n], dtype=torch.int32, device=dev)
    for up, mul in ((2.0, 1.0), (2.0 / 0.75, 0.75)):
        ho, wo = int(hm * up), int(wm * up)
        wo -= wo % 4
        masks = torch.zeros(B, n, ho, wo, dtype=torch.uint8, device=dev)
        H.mask_assemble(basis, False, cofs, keep, det, ndet, hm, wm, ho, wo, mul, 2.0, up, 0.4, masks)
        rect = torch.zeros(B * n, 4, dtype=torch.int32, device=dev)
        H.mask_rects(det, mul, 2.0, up, rect)
        r, m = rect.cpu().numpy(), masks.cpu().numpy()
        assert m.sum() > 0
        for i in range(n):
            ys, xs = np.nonzero(m[0, i])
            if len(ys):
                assert xs.min() >= r[i, 0] and xs.max() < r[i, 2] and ys.min() >= r[i, 1] and ys.max() < r[i, 3], (i, r[i])


@pytest.mark.parametrize("cfg", [(2, 64, 10, 12, 32, 1, 1), (1, 256, 13, 21, 256, 4, 1), (2, 128, 9, 7, 64, 1, 2),
                                 (1, 256, 40, 50, 256, 4, 1), (2, 64, 150, 120, 16, 1, 1)])   # last: 2 wgrad chunks
def test_deform_conv_backward_vs_oracle(cfg):
    """DeformConvFunction.backward (sm_deform_conv2d_bwd) against the oracle restatement of the reference's
    col2im / col2im_coord / parameter-gradient kernels on bf16-representable x, weight, grad_output.
    Tolerances (relative to the tensor's max): grad_input / grad_offset 6e-3 (the grad columns W^T gout are stored as bf16
    rows, like every other gradient of the row-tensor training graph -- 2^-9 per column element -- then f32 accumulation:
    LDS-window / HBM float atomics in any order; measured 2.1e-3), grad_weight 1e-2 (the sampled columns are rounded to
    bf16 before the MFMA, as in the forward)."""
    from sipmask_amd import ops as P
    dev = _dev()
    B, C, Hh, Ww, Co, G, dil = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = _bf(torch.randn(B, C, Hh, Ww, generator=g))
    w = _bf(torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    off = torch.randn(B, G * 18, Hh, Ww, generator=g) * 1.5
    off[:, :, 0, 0] = 0.0                                   # integer sampling points (lh = lw = 0)
    off[:, :, -1, -1] = 30.0                                # far outside: no contribution, zero offset gradient
    go = _bf(torch.randn(B, Co, Hh, Ww, generator=g))
    rx, roff, rw = O.deform_conv_backward(x, off, w, go, 1, dil, dil, G)
    xd, od, wd = x.to(dev).requires_grad_(), off.to(dev).requires_grad_(), w.to(dev).requires_grad_()
    y = P.deform_conv(xd, od, wd, 1, dil, dil, 1, G)
    y.backward(go.to(dev))
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / b.abs().max())
    assert rel(xd.grad, rx) < 6e-3, rel(xd.grad, rx)
    assert rel(od.grad, roff) < 6e-3, rel(od.grad, roff)
    assert rel(wd.grad, rw) < 1e-2, rel(wd.grad, rw)
    assert float(od.grad[:, :, -1, -1].abs().max()) == 0.0
    # only the weight gradient requested (backward_parameters alone, deform_conv.py:86-94)
    wd2 = w.to(dev).requires_grad_()
    P.deform_conv(x.to(dev), off.to(dev), wd2, 1, dil, dil, 1, G).backward(go.to(dev))
    assert rel(wd2.grad, rw) < 1e-2


@pytest.mark.parametrize("off_scale", [0.6, 2.5, 8.0])
def test_deform_dx_gather_equals_atomic_scatter(off_scale):
    """d(x) of FeatureAlign's deformable conv (round 5): samples within 3 pixels of their tap by the gather kernel (plain stores,
    fixed order), the farther ones by the atomic scatter behind it -- against the scatter alone (SM_CONV_BWD_DX_SCATTER, the
    kernel the oracle test above held until round 4) on a 3-level pyramid with tiles that hang over the image: equal up to
    f32 summation order, the offset gradients untouched, and two runs of the gather path bit-identical when no sample is far
    (0.6) -- the scatter's order is not reproducible.  8.0: nearly every sample takes the far path."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(41)
    B, C, Co, G = 2, 256, 256, 4
    sizes = [(21, 37), (11, 19), (5, 8)]
    lv = H.Levels(B, sizes)
    x = torch.randn(lv.rows, C, generator=g).to(torch.bfloat16).to(dev)
    off = (torch.randn(lv.rows, G * 18, generator=g) * off_scale).to(dev)
    go = torch.randn(lv.rows, Co, generator=g).to(torch.bfloat16).to(dev)
    w = torch.randn(Co, C, 3, 3, generator=g) / 48
    w_t, _ = H.prep_conv_weight(w.permute(2, 3, 1, 0).reshape(9 * C, Co, 1, 1).contiguous().to(dev), Co)

    def run(flags, want_goff=True):
        d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, C, Co, Co, 3, 1, 1, C, Co, flags=flags, deform_groups=G)
        gx = torch.full((lv.rows, C), float("nan"), dtype=torch.float32, device=dev)
        goff = torch.full_like(off, float("nan")) if want_goff else None
        H.deform_conv2d_bwd(d, x, off, w_t, go, gx, goff, None)
        torch.cuda.synchronize()
        return gx, goff

    a, ao = run(0)
    b, bo = run(L.SM_CONV_BWD_DX_SCATTER)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert torch.equal(ao, bo)
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2e-5 * scale, float((a - b).abs().max()) / scale
    # grad_x alone ("None = skip" for grad_offset): the far taps' d(x) comes from the scatter launch behind the
    # gather, which must run also when no offset gradient is asked for
    c, _ = run(0, want_goff=False)
    assert torch.isfinite(c).all()
    assert float((c - b).abs().max()) <= 2e-5 * scale, float((c - b).abs().max()) / scale
    if off_scale < 1.0:
        assert float((off.abs() > 3.0).float().sum()) == 0
        a2, _ = run(0)
        assert torch.equal(a, a2)


@pytest.mark.parametrize("cfg", [
    # (B, Cin, H, W, Cout, k, stride, pad, dil)
    (2, 64, 13, 17, 128, 3, 1, 1, 1),      # tower-like: fast dgrad (flipped-weight forward conv)
    (1, 256, 25, 42, 256, 3, 1, 1, 1),
    (2, 128, 12, 10, 64, 1, 1, 0, 1),      # 1x1
    (2, 64, 16, 20, 256, 1, 2, 0, 1),      # strided 1x1 (caffe-style conv1 / downsample): col2im dgrad
    (1, 256, 25, 42, 256, 3, 2, 1, 1),     # P6: stride-2 3x3
    (1, 64, 14, 14, 64, 3, 1, 2, 2),       # dilated
    (2, 64, 150, 120, 16, 3, 1, 1, 1),     # two wgrad chunks
])
def test_conv2d_backward_vs_torch(cfg):
    """sm_conv2d_bwd through ops.conv2d (autograd) against torch's CPU conv backward on bf16-representable x, w,
    grad_output: grad_input / grad_bias 2e-3, grad_weight 1e-2 of the tensor max (im2col operand is bf16)."""
    from sipmask_amd import ops as P
    dev = _dev()
    B, C, Hh, Ww, Co, k, s_, p_, dl = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = _bf(torch.randn(B, C, Hh, Ww, generator=g)).requires_grad_()
    w = _bf(torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5).requires_grad_()
    bias = torch.randn(Co, generator=g).requires_grad_()
    y = F.conv2d(x, w, bias, s_, p_, dl)
    go = _bf(torch.randn(y.shape, generator=g))
    y.backward(go)
    xd, wd, bd = (t.detach().to(dev).requires_grad_() for t in (x, w, bias))
    yd = P.conv2d(xd, wd, bd, s_, p_, dl)
    torch.testing.assert_close(yd.detach().cpu(), y.detach(), rtol=1e-4, atol=2e-4)
    yd.backward(go.to(dev))

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / b.abs().max())
    # strided convs take the grad-column path, whose columns are bf16 rows (2^-9 per element; measured 2.1e-3)
    assert rel(xd.grad, x.grad) < (6e-3 if s_ != 1 else 2e-3), rel(xd.grad, x.grad)
    assert rel(bd.grad, bias.grad) < 2e-3, rel(bd.grad, bias.grad)
    assert rel(wd.grad, w.grad) < 1e-2, rel(wd.grad, w.grad)


def test_training_layers_vs_torch():
    """ops.group_norm (+ReLU) and ops.upsample_bilinear, forward and backward, against torch CPU autograd (f32:
    1e-4 relative to the tensor max), and sm_sgd_step against the torch.optim.SGD update rule."""
    from sipmask_amd import ops as P, hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    for relu in (True, False):
        x = torch.randn(2, 64, 9, 11, generator=g).requires_grad_()
        w = (torch.rand(64, generator=g) + 0.5).requires_grad_()
        b = torch.randn(64, generator=g).requires_grad_()
        y = F.group_norm(x, 32, w, b, 1e-5)
        y = F.relu(y) if relu else y
        go = torch.randn(y.shape, generator=g)
        y.backward(go)
        xd, wd, bd = (t.detach().to(dev).requires_grad_() for t in (x, w, b))
        yd = P.group_norm(xd, wd, bd, 32, 1e-5, relu)
        torch.testing.assert_close(yd.detach().cpu(), y.detach(), rtol=1e-4, atol=1e-5)
        yd.backward(go.to(dev))
        for a, r in ((xd.grad, x.grad), (wd.grad, w.grad), (bd.grad, b.grad)):
            assert float((a.cpu() - r).abs().max()) <= 1e-4 * float(r.abs().max()) + 1e-6
    for f in (2, 4):
        x = torch.randn(2, 5, 7, 9, generator=g).requires_grad_()
        y = F.interpolate(x, scale_factor=f, mode="bilinear", align_corners=False)
        go = torch.randn(y.shape, generator=g)
        y.backward(go)
        xd = x.detach().to(dev).requires_grad_()
        yd = P.upsample_bilinear(xd, f)
        torch.testing.assert_close(yd.detach().cpu(), y.detach(), rtol=1e-5, atol=1e-6)
        yd.backward(go.to(dev))
        torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-4, atol=1e-5)
    # SGD with momentum + weight decay, two steps
    p0 = torch.randn(1000, generator=g)
    pr = p0.clone().requires_grad_()
    opt = torch.optim.SGD([pr], lr=0.01, momentum=0.9, weight_decay=1e-4)
    pd, buf = p0.clone().to(dev), torch.zeros(1000, device=dev)
    for step in range(2):
        gr = torch.randn(1000, generator=g)
        pr.grad = gr.clone()
        opt.step()
        H.sgd_step(pd, gr.to(dev), buf, 0.01, 0.9, 1e-4, step == 0)
    torch.testing.assert_close(pd.cpu(), pr.detach(), rtol=1e-6, atol=1e-7)


def test_input_pipeline_vs_oracle():
    """sm_preprocess_u8 (resize keep-ratio + normalise + pad + CHW) == the numpy restatement: identical except
    where the float interpolation lands within rounding distance of .5 (<= 1 grey level on < 0.1 % of the pixels).
    PARITY UNPINNED: cv2 / mmcv are absent, the oracle restates INTER_LINEAR's geometry in float; OpenCV's
    8-bit path is fixed point (oracle.pipeline.resize_bilinear_u8_fixedpoint restates it from OpenCV's source): the HIP
    output is held to that one too -- never more than ONE grey level away, the stated deviation bound of this row
    (tests/test_oracle_ops.py measures float restatement vs fixed point: <= 1 level on 5-13 % of the pixels)."""
    from sipmask_amd.input_pipeline import prepare_batch
    from oracle import pipeline as OP
    dev = _dev()
    rng = np.random.RandomState(3)
    imgs = [rng.randint(0, 256, (480, 640, 3)).astype(np.uint8), rng.randint(0, 256, (375, 500, 3)).astype(np.uint8),
            rng.randint(0, 256, (900, 1400, 3)).astype(np.uint8)]
    # smooth content as well (noise exercises the rounding edge, gradients the geometry)
    yy, xx = np.mgrid[:480, :640]
    imgs[0][..., 1] = ((yy * 0.3 + xx * 0.2) % 256).astype(np.uint8)
    batch, metas = prepare_batch([torch.from_numpy(i).to(dev) for i in imgs], img_scale=(1333, 800))
    hp, wp = batch.shape[-2:]
    assert hp % 32 == 0 and wp % 32 == 0
    for i, im in enumerate(imgs):
        ref, meta = OP.prepare(im, (1333, 800))
        assert metas[i]["img_shape"] == meta["img_shape"] and metas[i]["ori_shape"] == meta["ori_shape"]
        assert abs(metas[i]["scale_factor"] - meta["scale_factor"]) < 1e-12
        nh, nw = meta["img_shape"][:2]
        got = batch[i].cpu().numpy()
        d = np.abs(got[:, :nh, :nw] - ref[:, :nh, :nw])
        assert d.max() <= 1.0 + 1e-4 and (d > 1e-3).mean() < 1e-3, (d.max(), (d > 1e-3).mean())
        assert np.abs(got[:, nh:, :]).sum() == 0 and np.abs(got[:, :, nw:]).sum() == 0       # Pad with zeros
        fx = OP.resize_bilinear_u8_fixedpoint(im, nh, nw).astype(np.float32) - np.array([102.9801, 115.9465, 122.7717], np.float32)
        dfx = np.abs(got[:, :nh, :nw] - fx.transpose(2, 0, 1))
        assert dfx.max() <= 1.0 + 1e-4 and (dfx > 1e-3).mean() < 0.16, (dfx.max(), (dfx > 1e-3).mean())
    # SSD-style keep_ratio=False: scale factors [w, h, w, h]
    b2, m2 = prepare_batch([torch.from_numpy(imgs[1]).to(dev)], img_scale=(544, 544), keep_ratio=False)
    assert tuple(b2.shape) == (1, 3, 544, 544) and m2[0]["scale_factor"].shape == (4,)
    r2 = OP.resize_bilinear_u8(imgs[1], 544, 544).astype(np.float32) - np.array([102.9801, 115.9465, 122.7717], np.float32)
    d = np.abs(b2[0].cpu().numpy() - r2.transpose(2, 0, 1))
    assert d.max() <= 1.0 + 1e-4 and (d > 1e-3).mean() < 1e-3


def test_copy_segments_to_pinned_host_memory():
    """sm_copy_segments (round 6): several byte ranges -- aligned and unaligned, empty, 1 byte to 3 MB -- copied by ONE launch
    from device tensors into PINNED host buffers (and into device buffers); the host sees every byte once the stream has
    completed.  This is how a step's results leave the device (PipelinedPlan._pack, SipMaskVIS.clip_test_many)."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    sizes = [4000, 17, 1, 3 * 1024 * 1024 + 5, 400, 8 * 100 * 5 * 4]
    srcs = [torch.randint(0, 256, (n,), generator=g, dtype=torch.uint8).to(dev) for n in sizes]
    pinned = [torch.full((n,), 7, dtype=torch.uint8).pin_memory() for n in sizes]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        H.copy_segments(list(zip(srcs, pinned)))
    st.synchronize()
    for s, d in zip(srcs, pinned):
        assert torch.equal(s.cpu(), d)
    # unaligned source view, typed tensors, device destinations
    a = torch.arange(1000, dtype=torch.int32, device=dev)
    f = torch.randn(33, 5, generator=g).to(dev)
    da = torch.zeros(999, dtype=torch.int32).pin_memory()
    df = torch.zeros(33, 5, device=dev)
    H.copy_segments([(a[1:], da), (f, df)])
    torch.cuda.synchronize()
    assert torch.equal(da, a[1:].cpu()) and torch.equal(df, f)
    with pytest.raises(ValueError):
        H.copy_segments([(a, torch.zeros(999, dtype=torch.int32).pin_memory())])          # size mismatch
    with pytest.raises(ValueError):
        H.copy_segments([(a, torch.zeros(1000, dtype=torch.int32))])                      # pageable destination


def test_det_select_vs_oracle():
    """score -> per-level top-k -> gather/decode (sipmask_head.py:563-591) for a 2-image batch."""
    from sipmask_amd import hip_ops as H
    from oracle import model as OM
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    B, C = 2, 80
    sizes = [(40, 50), (20, 25), (10, 13), (5, 7), (3, 4)]
    strides = (8, 16, 32, 64, 128)
    lv = H.Levels(B, sizes)
    cls = [torch.randn(B, C, h, w, generator=g) * 2 - 3 for h, w in sizes]
    bbp = [torch.randn(B, 4, h, w, generator=g) * 2 + 3 for h, w in sizes]      # x Scale, before x stride
    ctr = [torch.randn(B, 1, h, w, generator=g) for h, w in sizes]
    cof = [torch.randn(B, 128, h, w, generator=g) for h, w in sizes]
    rows = lambda ts: torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts])
    cls_cof = torch.cat([rows(cls), rows(cof)], 1).contiguous().to(dev)
    reg = torch.cat([rows(bbp), rows(ctr), torch.zeros(lv.rows, 3)], 1).contiguous().to(dev)
    img_h, img_w = 320, 400
    d = H.make_det_desc(B, sizes, strides, lv.row0, C, C + 128, 0, C + 128, C, 8, 1000, img_h, img_w)
    out = H.det_select_alloc(d, dev)
    H.det_select(d, cls_cof, reg, cls_cof, out)
    torch.cuda.synchronize()
    for b in range(B):
        mb, ms, mc, mf, lvl, pos = OM.select_candidates([c[b] for c in cls], [x[b] * s for x, s in zip(bbp, strides)],
                                                        [c[b] for c in ctr], [c[b] for c in cof], (img_h, img_w, 3), 1000,
                                                        strides)
        assert int(out["ncand"][b]) == mb.shape[0] == d.kmax
        # EXACT: the ranking sigmoid is evaluated in double and rounded once on both sides (common.h sigmoid_rank ==
        # oracle.ops.sigmoid_ref), so candidate order, scores and centerness are the same f32 bits
        np.testing.assert_array_equal(out["cand_pos"][b].cpu().numpy(), pos.int().numpy())
        torch.testing.assert_close(out["boxes"][b].cpu(), mb, rtol=0, atol=0)
        torch.testing.assert_close(out["scores"][b].cpu().t(), ms[:, 1:], rtol=0, atol=0)
        torch.testing.assert_close(out["ctr"][b].cpu(), mc, rtol=0, atol=0)
        torch.testing.assert_close(out["cofs"][b].cpu(), mf, rtol=0, atol=0)


def test_mask_assemble_vs_oracle():
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    B, Hm, Wm, kmax, max_num = 2, 52, 76, 40, 12
    basis = torch.randn(B, 32, Hm, Wm, generator=g)
    cofs = torch.randn(B, kmax, 128, generator=g) * 0.4
    nd = [9, 0]
    det = torch.zeros(B, max_num, 5)
    xy = torch.rand(nd[0], 2, generator=g) * torch.tensor([2.0 * Wm * 0.7, 2.0 * Hm * 0.7])
    wh = torch.rand(nd[0], 2, generator=g) * 60 + 3
    det[0, :nd[0], :2] = xy
    det[0, :nd[0], 2:4] = xy + wh
    det[0, 0, :4] = torch.tensor([0.0, 0.0, 2.0 * Wm - 1, 2.0 * Hm - 1])     # full image
    det[0, 1, :4] = torch.tensor([10.3, 7.7, 10.9, 8.2])                      # sub-pixel box
    keep = torch.randint(0, kmax, (B, max_num), generator=g)
    for hwc in (True, False):
        bas = (basis.permute(0, 2, 3, 1) if hwc else basis).contiguous().to(dev)
        masks = torch.full((B, max_num, 2 * Hm, 2 * Wm), 7, dtype=torch.uint8, device=dev)
        posm = torch.full((B, max_num, Hm, Wm), -1.0, device=dev)
        H.mask_assemble(bas, hwc, cofs.to(dev), keep.to(dev), det.to(dev), torch.tensor(nd, dtype=torch.int32, device=dev),
                        Hm, Wm, 2 * Hm, 2 * Wm, 1.0, 2.0, 2.0, 0.4, masks, posm)
        torch.cuda.synchronize()
        r = O.mask_assemble(basis[0], cofs[0][keep[0, :nd[0]]], det[0, :nd[0]], 1.0, False)
        got_p = posm[0, :nd[0]].cpu()
        # mask "logits"/probabilities: stated tolerance 1e-3 (north_star); achieved ~1e-6
        torch.testing.assert_close(got_p, r["pos_masks"], rtol=0, atol=1e-5)
        assert bool(((got_p == 0) == (r["pos_masks"] == 0)).all())          # crop support identical
        gm = masks[0, :nd[0]].cpu()
        diff = gm != r["masks"]
        assert bool(((r["up"] - 0.4).abs()[diff] < 1e-5).all())             # only threshold-adjacent pixels may flip
        assert int(diff.sum()) <= 5
        assert bool((masks[0, nd[0]:] == 7).all()) and bool((masks[1] == 7).all())   # rows >= ndet untouched


def test_mask_assemble_lo_vs_oracle_and_rectangle_tracking():
    """sm_mask_assemble_lo: basis at conv resolution (the x4 bilinear of sipmask_head.py:285 applied AFTER the coefficient
    dot product), only the detections' rectangles written, the previous call's rectangles cleared.  Against the oracle
    fed with F.interpolate(basis_lo, x4): masks may differ only where the upsampled probability is within 1e-5 of
    the threshold (the reassociation changes logits by f32 rounding), and after every call the WHOLE buffer -- slots
    beyond ndet, pixels outside the boxes, rectangles of the previous call -- must be what the oracle says."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(33)
    B, h0, w0, kmax, max_num = 2, 13, 21, 60, 9
    Hm, Wm = 4 * h0, 4 * w0
    Ho, Wo = 2 * Hm, 2 * Wm
    basis_lo = torch.relu(torch.randn(B, 32, h0, w0, generator=g))
    feat = F.interpolate(basis_lo, scale_factor=4, mode="bilinear", align_corners=False)
    cofs = torch.randn(B, kmax, 128, generator=g) * 0.5
    lo_rows = basis_lo.permute(0, 2, 3, 1).contiguous().to(dev)
    buf = H.mask_assemble_lo_alloc(B, max_num, Ho, Wo, dev)
    for it, nd in enumerate(([7, 3], [2, 9], [0, 1])):                 # detections move, counts shrink and grow
        xy = torch.rand(B, max_num, 2, generator=g) * torch.tensor([Wo * 0.7, Ho * 0.7])
        wh = torch.rand(B, max_num, 2, generator=g) * torch.tensor([Wo * 0.5, Ho * 0.5]) + 3
        det = torch.cat([xy, xy + wh, torch.rand(B, max_num, 1, generator=g)], 2)
        if it == 0:
            det[0, 0, :4] = torch.tensor([-20.0, -10.0, Wo + 30.0, Ho + 5.0])         # larger than the image
            det[0, 1, :4] = torch.tensor([10.3, 7.7, 10.9, 8.2])                      # sub-pixel box
        keep = torch.randint(0, kmax, (B, max_num), generator=g)
        H.mask_assemble_lo(lo_rows, h0, w0, 4, cofs.to(dev), keep.to(dev), det.to(dev),
                           torch.tensor(nd, dtype=torch.int32, device=dev), Ho, Wo, 1.0, 2.0, 2.0, 0.4, buf)
        torch.cuda.synchronize()
        got = buf["masks"][..., :Wo].cpu()
        for b in range(B):
            n = nd[b]
            assert int(got[b, n:].sum()) == 0                          # stale slots cleared
            if n == 0:
                continue
            r = O.mask_assemble(feat[b], cofs[b][keep[b, :n]], det[b, :n], 1.0, False)
            diff = got[b, :n] != r["masks"]
            assert bool(((r["up"] - 0.4).abs()[diff] < 1e-5).all()), (it, b)
            assert int(diff.sum()) <= 5


@pytest.mark.parametrize("sf", [1.667, 2.0, 2.7, 0.8])
def test_mask_assemble_lo_keep_ratio_scale_factors(sf):
    """get_bboxes upsamples by 2 / scale_factor (sipmask_head.py:632), and a keep_ratio COCO resize gives
    scale factors of 1.6-2.7, i.e. up_scale down to 0.74 -- a 128x8 output tile then reads a source window several times
    the one of up_scale 2.  The kernel sizes its LDS tiles per launch; every such geometry must be supported and match the
    oracle (crop boxes scaled by scale_factor / 2, :623), and the plan-build query must say so."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(int(sf * 1000))
    B, h0, w0, kmax, max_num = 2, 13, 21, 40, 6
    Hm, Wm = 4 * h0, 4 * w0
    mul, up, (Ho, Wo) = H.post_geometry(Hm, Wm, sf, True)
    assert H.mask_assemble_lo_supported(B, max_num, 4, up)
    assert not H.mask_assemble_lo_supported(B, max_num, 4, 2.0 / 8.0)          # (far outside any pipeline: falls back)
    basis_lo = torch.relu(torch.randn(B, 32, h0, w0, generator=g))
    feat = F.interpolate(basis_lo, scale_factor=4, mode="bilinear", align_corners=False)
    cofs = torch.randn(B, kmax, 128, generator=g) * 0.5
    lo_rows = basis_lo.permute(0, 2, 3, 1).contiguous().to(dev)
    buf = H.mask_assemble_lo_alloc(B, max_num, Ho, Wo, dev)
    for nd in ([6, 2], [1, 5]):
        # boxes in ORIGINAL-image coordinates (rescale=True: det / scale_factor), so that box * sf / 2 lands on the grid
        xy = torch.rand(B, max_num, 2, generator=g) * torch.tensor([Wo * 0.7, Ho * 0.7])
        wh = torch.rand(B, max_num, 2, generator=g) * torch.tensor([Wo * 0.5, Ho * 0.5]) + 3
        det = torch.cat([xy, xy + wh, torch.rand(B, max_num, 1, generator=g)], 2)
        keep = torch.randint(0, kmax, (B, max_num), generator=g)
        H.mask_assemble_lo(lo_rows, h0, w0, 4, cofs.to(dev), keep.to(dev), det.to(dev),
                           torch.tensor(nd, dtype=torch.int32, device=dev), Ho, Wo, mul, 2.0, up, 0.4, buf)
        torch.cuda.synchronize()
        got = buf["masks"][..., :Wo].cpu()
        for b in range(B):
            n = nd[b]
            assert int(got[b, n:].sum()) == 0
            r = O.mask_assemble(feat[b], cofs[b][keep[b, :n]], det[b, :n], sf, True)
            assert tuple(r["masks"].shape[-2:]) == (Ho, Wo)
            diff = got[b, :n] != r["masks"]
            assert bool(((r["up"] - 0.4).abs()[diff] < 1e-5).all()), (sf, b)
            assert int(diff.sum()) <= 5


def test_crop_split_and_gt_vs_oracle():
    from sipmask_amd import ops as P
    dev = _dev()
    rng = np.random.RandomState(0)
    Hh, Ww, N = 37, 45, 11
    data = rng.rand(4, Hh, Ww, N).astype(np.float32)
    xy = rng.rand(N, 2).astype(np.float32) * 30 - 5
    rois = np.concatenate([xy, xy + rng.rand(N, 2).astype(np.float32) * 40 + 0.05], 1).astype(np.float32)
    t = torch.from_numpy(data).to(dev).requires_grad_(True)
    r = torch.from_numpy(rois).to(dev)
    out = P.CropSplit(2)(t, r)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.crop_split(data, rois, 2))
    go = rng.rand(Hh, Ww, N).astype(np.float32)
    out.backward(torch.from_numpy(go).to(dev))
    np.testing.assert_array_equal(t.grad.cpu().numpy(), O.crop_split_backward(go, rois, 2))
    gt = P.CropSplitGt(2)(torch.from_numpy(data[0]).to(dev), r)
    np.testing.assert_array_equal(gt.cpu().numpy(), O.crop_split_gt(data[0], rois))
    with pytest.raises(NotImplementedError):
        P.CropSplit(2)(torch.from_numpy(data), torch.from_numpy(rois))      # CPU tensors: no CPU path


def test_sigmoid_focal_loss_vs_oracle():
    from sipmask_amd import ops as P
    dev = _dev()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(777, 80, generator=g) * 4
    t = torch.randint(0, 81, (777,), generator=g)
    xr = x.clone().to(dev).requires_grad_(True)
    loss = P.sigmoid_focal_loss(xr, t.to(dev), 2.0, 0.25)
    ref = O.sigmoid_focal_loss_forward(x.double(), t, 2.0, 0.25)
    torch.testing.assert_close(loss.detach().cpu().double(), ref, rtol=2e-5, atol=1e-7)
    dl = torch.rand(777, 80, generator=g)
    loss.backward(dl.to(dev))
    refg = O.sigmoid_focal_loss_backward(x.double(), t, dl.double(), 2.0, 0.25)
    torch.testing.assert_close(xr.grad.cpu().double(), refg, rtol=2e-5, atol=1e-7)


def test_relu_bf16_and_p7_on_the_dma_path():
    """sm_relu_bf16 is exact relu on bf16 (sign bit always cleared: -0 -> +0), and the P7 conv fed with relu(P6) (LDS-DMA operand
    path, split-K) equals the same conv with the input-ReLU flag (register-staged loader): fpn.py:166-170."""
    from sipmask_amd import hip_ops as H
    from sipmask_amd._lib import SM_CONV_IN_RELU
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2 * 13 * 21, 256, generator=g).to(torch.bfloat16)
    x[0, :8] = torch.tensor([-0.0, 0.0, -1.0, 1.0, -3e-38, 3e-38, -65280.0, 65280.0], dtype=torch.bfloat16)
    xd = x.to(dev)
    y = H.relu_bf16(xd, torch.empty_like(xd))
    yc = y.cpu()
    assert torch.equal(yc.float(), torch.relu(x.float())) and not bool((yc.view(torch.int16) < 0).any())   # -0 -> +0
    w = (torch.randn(256, 256, 3, 3, generator=g) / 48).to(torch.bfloat16).float()
    wp, cop = H.prep_conv_weight(w.to(dev), 256)
    outs = []
    for flags, src in ((SM_CONV_IN_RELU, xd), (0, y)):
        d = H.make_conv_desc(2, [(13, 21)], [(7, 11)], [0], [0], 256, 256, cop, 3, 2, 1, 256, 256, 0, flags)
        o = torch.zeros(2 * 7 * 11, 256, dtype=torch.bfloat16, device=dev)
        pl = H.conv_plan(d)
        if pl["split_k"] > 1:
            H.conv2d_ws(d, src, wp, None, None, o, torch.empty(pl["workspace_bytes"], dtype=torch.uint8, device=dev))
        else:
            H.conv2d(d, src, wp, None, None, o)
        outs.append(o.float().cpu())
    ref = torch.nn.functional.conv2d(torch.relu(x.float()).view(2, 13, 21, 256).permute(0, 3, 1, 2), w, None, 2, 1)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, 256)
    for o in outs:
        torch.testing.assert_close(o, ref, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(outs[0], outs[1], rtol=0, atol=2 ** -6)


@pytest.mark.parametrize("C,chain", [(64, False), (64, True), (128, False), (128, True)])
def test_bottleneck_tail_bit_identical_to_separate_convs(C, chain):
    """sm_bottleneck_tail (conv2 3x3 + conv3 1x1 + identity [+ the next block's conv1] in one launch; resnet.py:167-200)
    against the same three convs as separate sm_conv2d launches: same K order and rounding points -> identical bf16
    bits; against torch fp32 on the bf16-rounded intermediates -> accumulation-order tolerance.  M = 2*13*19 = 494 rows
    (not a multiple of the 128-position tile), borders everywhere."""
    from sipmask_amd import hip_ops as H
    from sipmask_amd._lib import SM_CONV_RELU, SM_CONV_RES_ADD
    dev = _dev()
    B, h, w = 2, 13, 19
    M = B * h * w
    g = torch.Generator().manual_seed(C + chain)
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(torch.randn(M, C, generator=g)).to(dev)
    idt = bf(torch.randn(M, 4 * C, generator=g)).to(dev)
    w2 = bf(torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).float()
    w3 = bf(torch.randn(4 * C, C, 1, 1, generator=g) / C ** 0.5).float()
    w1 = bf(torch.randn(C, 4 * C, 1, 1, generator=g) / (2 * C ** 0.5)).float()
    b2, b3, b1 = (torch.randn(n, generator=g).to(dev) * 0.1 for n in (C, 4 * C, C))
    # separate launches
    def conv(src, wt, bias, cin, cout, k, flags, res=None):
        wp, cop = H.prep_conv_weight(wt.to(dev), cin)
        d = H.make_conv_desc(B, [(h, w)], [(h, w)], [0], [0], cin, cout, cop, k, 1, k // 2, cin, cout, 0, flags, 1,
                             cout if res is not None else 0)
        o = torch.zeros(M, cout, dtype=torch.bfloat16, device=dev)
        H.conv2d(d, src, wp, bias, res, o)
        return o
    t2 = conv(x, w2, b2, C, C, 3, SM_CONV_RELU)
    y_ref = conv(t2, w3, b3, C, 4 * C, 1, SM_CONV_RELU | SM_CONV_RES_ADD, idt)
    t1_ref = conv(y_ref, w1, b1, 4 * C, C, 1, SM_CONV_RELU)
    # fused
    prep = lambda wt: H.prep_conv_weight(wt.to(dev), wt.shape[1])[0][:wt.shape[0]].contiguous()
    y = torch.zeros(M + 7, 4 * C, dtype=torch.bfloat16, device=dev)
    t1n = torch.zeros(M + 7, C, dtype=torch.bfloat16, device=dev)
    H.bottleneck_tail(B, h, w, C, x, prep(w2), b2, prep(w3), b3, idt, y, *( (prep(w1), b1, t1n) if chain else ()))
    torch.cuda.synchronize()
    assert torch.equal(y[:M].view(torch.int16), y_ref.view(torch.int16))
    assert not bool(y[M:].any()) and not bool(t1n[M:].any())            # nothing written past M
    if chain:
        assert torch.equal(t1n[:M].view(torch.int16), t1_ref.view(torch.int16))
    # torch fp32 on the same rounding points
    xi = x.float().cpu().view(B, h, w, C).permute(0, 3, 1, 2)
    r2 = bf(torch.relu(torch.nn.functional.conv2d(xi, w2, b2.cpu(), 1, 1))).float()
    r3 = torch.relu(torch.nn.functional.conv2d(r2, w3, b3.cpu()) + idt.float().cpu().view(B, h, w, 4 * C).permute(0, 3, 1, 2))
    torch.testing.assert_close(y[:M].float().cpu(), r3.permute(0, 2, 3, 1).reshape(M, 4 * C), rtol=1e-2, atol=2e-2)


def test_bottleneck_tail_with_fused_shortcut_conv():
    """sm_bottleneck_tail_ds (round 4): conv2 3x3 + conv3 1x1 + the block's 1x1 SHORTCUT conv (resnet.py:453-469; layer1's
    first block) in one launch -- the shortcut's K rides conv3's accumulator, [w3 | w_downsample] rows.  Against torch fp32 on
    the same bf16 rounding points (conv2 output rounded; shortcut NOT rounded: it stays f32 until the block output's
    rounding), and against the two-launch path (separate shortcut conv, rounded to bf16, + sm_bottleneck_tail): equal up to
    that one extra rounding.  M = 2*13*19 rows (not a multiple of the 128-position tile), borders everywhere."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    B, h, w, C = 2, 13, 19, 64
    M = B * h * w
    g = torch.Generator().manual_seed(11)
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(torch.randn(M, C, generator=g)).to(dev)                    # conv1 output (the tail's input)
    xb = bf(torch.relu(torch.randn(M, 64, generator=g))).to(dev)      # block input (max-pool output: non-negative)
    w2 = bf(torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).float()
    w3 = bf(torch.randn(4 * C, C, 1, 1, generator=g) / C ** 0.5).float()
    wd = bf(torch.randn(4 * C, 64, 1, 1, generator=g) / 8.0).float()
    b2, b3, bd = (torch.randn(n, generator=g).to(dev) * 0.1 for n in (C, 4 * C, 4 * C))
    prep = lambda wt: H.prep_conv_weight(wt.to(dev), wt.shape[1])[0][:wt.shape[0]].contiguous()
    w3ds = torch.cat([prep(w3), prep(wd)], 1).contiguous()
    assert tuple(w3ds.shape) == (4 * C, 128)
    y = torch.zeros(M + 7, 4 * C, dtype=torch.bfloat16, device=dev)
    H.bottleneck_tail_ds(B, h, w, C, x, prep(w2), b2, w3ds, (b3 + bd).contiguous(), xb, y)
    torch.cuda.synchronize()
    assert not bool(y[M:].any())                                       # nothing written past M
    nchw = lambda t, c: t.float().cpu().view(B, h, w, c).permute(0, 3, 1, 2)
    r2 = bf(torch.relu(F.conv2d(nchw(x, C), w2, b2.cpu(), 1, 1))).float()
    sc = F.conv2d(nchw(xb, 64), wd, bd.cpu())
    r3 = torch.relu(F.conv2d(r2, w3, b3.cpu()) + sc)
    ref = bf(r3).float().permute(0, 2, 3, 1).reshape(M, 4 * C)
    got = y[:M].float().cpu()
    diff = (got - ref).abs()
    assert bool((diff <= 2.0 ** -6 * ref.abs() + 1e-3).all()), float(diff.max())     # one bf16 ulp (f32 summation order; 2^-7 of
    assert float((diff > 0).float().mean()) < 0.05                                    # the value just above a power of two)
    # two-launch path: shortcut conv as its own launch (rounded to bf16), then the tail with it as identity
    dsc = H.make_conv_desc(B, [(h, w)], [(h, w)], [0], [0], 64, 4 * C, 4 * C, 1, 1, 0, 64, 4 * C)
    idt = torch.zeros(M, 4 * C, dtype=torch.bfloat16, device=dev)
    H.conv2d(dsc, xb, H.prep_conv_weight(wd.to(dev), 64)[0], bd, None, idt)
    y2 = torch.zeros(M, 4 * C, dtype=torch.bfloat16, device=dev)
    H.bottleneck_tail(B, h, w, C, x, prep(w2), b2, prep(w3), b3, idt, y2)
    torch.testing.assert_close(got, y2.float().cpu(), rtol=2 ** -6, atol=2e-2)
    # ... and with the next block's conv1 chained behind it: same block output bits, t1 = the separate conv1 launch on it
    from sipmask_amd._lib import SM_CONV_RELU
    w1 = bf(torch.randn(C, 4 * C, 1, 1, generator=g) / (2 * C ** 0.5)).float()
    b1 = (torch.randn(C, generator=g) * 0.1).to(dev)
    y3 = torch.zeros(M + 7, 4 * C, dtype=torch.bfloat16, device=dev)
    t1n = torch.zeros(M + 7, C, dtype=torch.bfloat16, device=dev)
    H.bottleneck_tail_ds(B, h, w, C, x, prep(w2), b2, w3ds, (b3 + bd).contiguous(), xb, y3, prep(w1), b1, t1n)
    torch.cuda.synchronize()
    assert torch.equal(y3.view(torch.int16), y.view(torch.int16)) and not bool(t1n[M:].any())
    d1 = H.make_conv_desc(B, [(h, w)], [(h, w)], [0], [0], 4 * C, C, C, 1, 1, 0, 4 * C, C, 0, SM_CONV_RELU)
    t1_ref = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
    H.conv2d(d1, y[:M].contiguous(), H.prep_conv_weight(w1.to(dev), 4 * C)[0], b1, None, t1_ref)
    assert torch.equal(t1n[:M].view(torch.int16), t1_ref.view(torch.int16))


@pytest.mark.parametrize("ds_hw", [(26, 38), (25, 37)])
def test_bottleneck_tail_with_fused_stride2_shortcut(ds_hw):
    """sm_bottleneck_tail_ds for layer2's first block: 128 channels, the shortcut a 1x1 / stride-2 conv 256 -> 512 of the block
    input on the (2h, 2w) or (2h - 1, 2w - 1) grid (resnet.py:453-469) -- output (n, ho, wo) reads input (n, 2 ho, 2 wo).
    Against torch fp32 on the same bf16 rounding points."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    B, h, w, C, Cd = 2, 13, 19, 128, 256
    assert ((ds_hw[0] - 1) // 2 + 1, (ds_hw[1] - 1) // 2 + 1) == (h, w)
    M = B * h * w
    g = torch.Generator().manual_seed(ds_hw[0])
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(torch.randn(M, C, generator=g)).to(dev)
    xb = bf(torch.relu(torch.randn(B * ds_hw[0] * ds_hw[1], Cd, generator=g))).to(dev)
    w2 = bf(torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).float()
    w3 = bf(torch.randn(4 * C, C, 1, 1, generator=g) / C ** 0.5).float()
    wd = bf(torch.randn(4 * C, Cd, 1, 1, generator=g) / 16.0).float()
    b2, b3, bd = (torch.randn(n, generator=g).to(dev) * 0.1 for n in (C, 4 * C, 4 * C))
    prep = lambda wt: H.prep_conv_weight(wt.to(dev), wt.shape[1])[0][:wt.shape[0]].contiguous()
    w3ds = torch.cat([prep(w3), prep(wd)], 1).contiguous()
    assert tuple(w3ds.shape) == (4 * C, C + Cd)
    y = torch.zeros(M + 7, 4 * C, dtype=torch.bfloat16, device=dev)
    H.bottleneck_tail_ds(B, h, w, C, x, prep(w2), b2, w3ds, (b3 + bd).contiguous(), xb, y, ds_stride=2, ds_hw=ds_hw)
    torch.cuda.synchronize()
    assert not bool(y[M:].any())
    r2 = bf(torch.relu(F.conv2d(x.float().cpu().view(B, h, w, C).permute(0, 3, 1, 2), w2, b2.cpu(), 1, 1))).float()
    sc = F.conv2d(xb.float().cpu().view(B, ds_hw[0], ds_hw[1], Cd).permute(0, 3, 1, 2), wd, bd.cpu(), 2)
    ref = bf(torch.relu(F.conv2d(r2, w3, b3.cpu()) + sc)).float().permute(0, 2, 3, 1).reshape(M, 4 * C)
    diff = (y[:M].float().cpu() - ref).abs()
    assert bool((diff <= 2.0 ** -6 * ref.abs() + 1e-3).all()), float(diff.max())
    assert float((diff > 0).float().mean()) < 0.05
    with pytest.raises(Exception):                 # the grid must match the stride
        H.bottleneck_tail_ds(B, h, w, C, x, prep(w2), b2, w3ds, b3, xb, y, ds_stride=2, ds_hw=(ds_hw[0] + 2, ds_hw[1]))


def test_conv1x1_pair_bit_identical_to_two_launches():
    """sm_conv1x1_pair (round 4; csrc/bottleneck.hip with CONV2 = false): conv3 (256 -> 1024, + identity, ReLU) of one layer3
    bottleneck and conv1 (1024 -> 256, ReLU) of the next in one launch (resnet.py:188-200, :175-178), against the same two
    convs as separate sm_conv2d launches: identical bf16 bits (same K order, same rounding points).  M = 2 * 13 * 19 rows (not a
    multiple of the 128-position tile)."""
    from sipmask_amd import hip_ops as H
    from sipmask_amd._lib import SM_CONV_RELU, SM_CONV_RES_ADD
    dev = _dev()
    B, h, w, C = 2, 13, 19, 256
    M = B * h * w
    g = torch.Generator().manual_seed(5)
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(torch.relu(torch.randn(M, C, generator=g))).to(dev)
    idt = bf(torch.randn(M, 4 * C, generator=g)).to(dev)
    w3 = bf(torch.randn(4 * C, C, 1, 1, generator=g) / C ** 0.5).float()
    w1 = bf(torch.randn(C, 4 * C, 1, 1, generator=g) / (2 * C ** 0.5)).float()
    b3, b1 = (torch.randn(n, generator=g).to(dev) * 0.1 for n in (4 * C, C))

    def conv(src, wt, bias, cin, cout, flags, res=None):
        wp, cop = H.prep_conv_weight(wt.to(dev), cin)
        d = H.make_conv_desc(B, [(h, w)], [(h, w)], [0], [0], cin, cout, cop, 1, 1, 0, cin, cout, 0, flags, 1,
                             cout if res is not None else 0)
        o = torch.zeros(M, cout, dtype=torch.bfloat16, device=dev)
        H.conv2d(d, src, wp, bias, res, o)
        return o
    y_ref = conv(x, w3, b3, C, 4 * C, SM_CONV_RELU | SM_CONV_RES_ADD, idt)
    t1_ref = conv(y_ref, w1, b1, 4 * C, C, SM_CONV_RELU)
    prep = lambda wt: H.prep_conv_weight(wt.to(dev), wt.shape[1])[0][:wt.shape[0]].contiguous()
    y = torch.zeros(M + 7, 4 * C, dtype=torch.bfloat16, device=dev)
    t1n = torch.zeros(M + 7, C, dtype=torch.bfloat16, device=dev)
    H.conv1x1_pair(M, C, x, prep(w3), b3, idt, y, prep(w1), b1, t1n)
    torch.cuda.synchronize()
    assert not bool(y[M:].any()) and not bool(t1n[M:].any())
    assert torch.equal(y[:M].view(torch.int16), y_ref.view(torch.int16))
    assert torch.equal(t1n[:M].view(torch.int16), t1_ref.view(torch.int16))
    r3 = torch.relu(x.float().cpu() @ w3.view(4 * C, C).t() + b3.cpu() + idt.float().cpu())
    torch.testing.assert_close(y[:M].float().cpu(), r3, rtol=1e-2, atol=2e-2)


def test_upsample_sum2_vs_torch_and_lat0_by_linearity():
    """sm_upsample_sum2 (round 4): out = [relu](a0 + up2(a1) + up4(a2)), bilinear align_corners=False, against
    F.interpolate -- bf16 rows (one rounding), f32 rows (1e-6), the split layout [hi | lo | hi] (hi + lo == f32 value to 2^-21) --
    and the identity the launch plan uses it for: the 1x1 conv sip_mask_lat0 over [l0 | up2(l1) | up4(l2)]
    (sipmask_head.py:275-283) equals W0.l0 + up2(W1.l1) + up4(W2.l2) up to f32 rounding."""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    B, h0, w0, C = 2, 12, 20, 32
    a0 = torch.randn(B, C, h0, w0, generator=g)
    a1 = torch.randn(B, C, h0 // 2, w0 // 2, generator=g)
    a2 = torch.randn(B, C, h0 // 4, w0 // 4, generator=g)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
    up = lambda t, f: F.interpolate(t, scale_factor=f, mode="bilinear", align_corners=False)
    # f32 rows, with and without a0 / relu
    want = a0 + up(a1, 2) + up(a2, 4)
    out = torch.empty(B * h0 * w0, C, device=dev)
    H.upsample_sum2(rows(a0).to(dev), rows(a1).to(dev), rows(a2).to(dev), out, B, h0, w0, C)
    torch.testing.assert_close(out.cpu(), rows(want), rtol=1e-6, atol=1e-6)
    H.upsample_sum2(None, rows(a1).to(dev), rows(a2).to(dev), out, B, h0, w0, C, relu=True)
    torch.testing.assert_close(out.cpu(), rows(torch.relu(up(a1, 2) + up(a2, 4))), rtol=1e-6, atol=1e-6)
    # split output
    o3 = torch.empty(B * h0 * w0, 3 * C, dtype=torch.float16, device=dev)
    H.upsample_sum2(rows(a0).to(dev), rows(a1).to(dev), rows(a2).to(dev), o3, B, h0, w0, C, relu=True)
    o3 = o3.cpu().float()
    assert torch.equal(o3[:, :C], o3[:, 2 * C:])
    torch.testing.assert_close(o3[:, :C] + o3[:, C:2 * C], rows(torch.relu(want)), rtol=2e-6, atol=2e-6)
    # bf16 rows
    b1, b2 = a1.to(torch.bfloat16), a2.to(torch.bfloat16)
    ob = torch.empty(B * h0 * w0, C, dtype=torch.bfloat16, device=dev)
    H.upsample_sum2(None, rows(b1).to(dev), rows(b2).to(dev), ob, B, h0, w0, C)
    torch.testing.assert_close(ob.cpu().float(), rows(up(b1.float(), 2) + up(b2.float(), 4)), rtol=2 ** -7, atol=1e-3)
    # the identity
    w = torch.randn(16, 3 * C, 1, 1, generator=g) / (3 * C) ** 0.5
    l0, l1, l2 = a0, a1, a2
    direct = F.conv2d(torch.cat([l0, up(l1, 2), up(l2, 4)], 1), w)
    by_lin = F.conv2d(l0, w[:, :C]) + up(F.conv2d(l1, w[:, C:2 * C]), 2) + up(F.conv2d(l2, w[:, 2 * C:]), 4)
    torch.testing.assert_close(by_lin, direct, rtol=1e-5, atol=1e-5)
