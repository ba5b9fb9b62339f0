"""GPU parity of the row-tensor training ops (sipmask_amd/ops_rows.py, csrc/train_rows.hip) against plain PyTorch fp32
autograd of the same op on the same bf16-representable inputs.  Forward outputs and activation gradients are bf16 rows
(one rounding: rtol 2^-7), parameter gradients f32 accumulated from bf16 operands (relative Frobenius error bounds).
The whole-graph wiring is covered by tests/test_gpu_api.py (head / detector training steps vs the oracle's autograd),
which run through these ops by default."""
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


def _rows(t):
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])


def _nchw(r, b, h, w):
    return r.float().view(b, h, w, -1).permute(0, 3, 1, 2)


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def test_weight_prep_layouts_match_the_tensor_code():
    """sm_weight_prep (one launch) == hip_ops.prep_conv_weight applied to w, flip+transpose(w), K-major(w), with and
    without the per-cout scale; sm_wgrad_finish == the permute + scale it replaces"""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(0)
    for co, ci, k, cin_pad in ((64, 3, 7, 8), (256, 64, 1, 64), (72, 128, 3, 128), (8, 256, 3, 256), (512, 2048, 1, 2048)):
        w = torch.randn(co, ci, k, k, generator=g).to(dev)
        s = (torch.rand(co, generator=g) + 0.5).to(dev)
        for scale in (None, s):
            ws = w if scale is None else w * scale.view(-1, 1, 1, 1)
            a, ra = H.weight_prep(w, scale, 0, cin_pad)
            b, rb = H.prep_conv_weight(ws, cin_pad)
            assert ra == rb and torch.equal(a, b)
            a, _ = H.weight_prep(w, scale, 1)
            b, _ = H.prep_conv_weight(ws.flip(2, 3).permute(1, 0, 2, 3).contiguous(), co)
            assert torch.equal(a, b)
            a, _ = H.weight_prep(w, scale, 2)
            b, _ = H.prep_conv_weight(ws.permute(2, 3, 1, 0).reshape(k * k * ci, co, 1, 1).contiguous(), co)
            assert torch.equal(a, b)
            gw_t = torch.randn(k * k * ci, co, generator=g).to(dev)
            ref = gw_t.view(k, k, ci, co).permute(3, 2, 0, 1) * (1.0 if scale is None else scale.view(-1, 1, 1, 1))
            torch.testing.assert_close(H.wgrad_finish(gw_t, scale, co, ci, k, k), ref.contiguous(), rtol=1e-6, atol=0)


def test_elementwise_and_resampling_adjoints():
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(1)
    # relu backward: gate on y > 0 (y a ReLU output incl. exact zeros and -0)
    y = torch.relu(torch.randn(1000, 64, generator=g)).to(torch.bfloat16)
    y[0, :8] = -0.0
    gg = torch.randn(1000, 64, generator=g).to(torch.bfloat16)
    out = H.relu_bwd_bf16(gg.to(dev), y.to(dev)).cpu()
    assert torch.equal(out, torch.where(y.float() > 0, gg, torch.zeros_like(gg)))
    # bias gradient = column sums
    for rows, c in ((5000, 256), (777, 8), (1, 208)):
        t = torch.randn(rows, c, generator=g).to(torch.bfloat16)
        torch.testing.assert_close(H.bias_grad_rows(t.to(dev), c).cpu(), t.float().sum(0), rtol=1e-4, atol=1e-2)
    # bilinear upsampling adjoint, reading a channel slice of a wider gradient
    for (b, h, w, c, f) in ((2, 5, 7, 16, 2), (1, 4, 5, 8, 4), (2, 13, 21, 256, 2)):
        x = torch.randn(b, c, h, w, generator=g, requires_grad=True)
        go = _bf(torch.randn(b, c, h * f, w * f, generator=g))
        F.interpolate(x, scale_factor=f, mode='bilinear', align_corners=False).backward(go)
        wide = torch.zeros(b * h * f * w * f, 3 * c, dtype=torch.bfloat16)
        wide[:, c:2 * c] = _rows(go).to(torch.bfloat16)
        got = H.upsample_bilinear_bwd_rows(wide.to(dev), 3 * c, c, b, h, w, c, f).cpu()
        torch.testing.assert_close(_nchw(got, b, h, w), x.grad, rtol=2 ** -7, atol=2e-2)
    # nearest-neighbour adjoint (FPN top-down), exact and ragged ratios
    for (b, hc, wc, hf, wf, c) in ((2, 4, 5, 8, 10, 16), (1, 13, 21, 25, 42, 8), (2, 7, 11, 13, 21, 8)):
        x = torch.randn(b, c, hc, wc, generator=g, requires_grad=True)
        go = _bf(torch.randn(b, c, hf, wf, generator=g))
        F.interpolate(x, size=(hf, wf), mode='nearest').backward(go)
        got = H.nearest_bwd_rows(_rows(go).to(torch.bfloat16).contiguous().to(dev), b, (hf, wf), (hc, wc), c).cpu()
        torch.testing.assert_close(_nchw(got, b, hc, wc), x.grad, rtol=2 ** -7, atol=2e-2)
    # strided scatter
    b, h, w, c, s = 2, 7, 10, 16, 2
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    t = torch.randn(b * ho * wo, c, generator=g).to(torch.bfloat16)
    got = H.scatter_stride_rows(t.to(dev), b, (h, w), (ho, wo), s, c).cpu().view(b, h, w, c)
    ref = torch.zeros(b, h, w, c, dtype=torch.bfloat16)
    ref[:, ::s, ::s] = t.view(b, ho, wo, c)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("relu", [True, False])
def test_gn_rows_forward_backward(relu):
    from sipmask_amd import hip_ops as H, ops_rows as R
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    b, c = 2, 256
    sizes = [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)]
    lv = H.Levels(b, sizes)
    xs = [_bf(torch.randn(b, c, h, w, generator=g) * 1.5 + 0.3).requires_grad_(True) for h, w in sizes]
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(c, generator=g) * 0.3).requires_grad_(True)
    gos = [_bf(torch.randn(b, c, h, w, generator=g)) for h, w in sizes]
    refs = []
    for x, go in zip(xs, gos):
        y = F.group_norm(x, 32, gamma, beta, 1e-5)
        y = F.relu(y) if relu else y
        y.backward(go)
        refs.append(y.detach())
    xr = torch.cat([_rows(x.detach()) for x in xs]).to(torch.bfloat16).to(dev).requires_grad_(True)
    gm, bt = gamma.detach().to(dev).requires_grad_(True), beta.detach().to(dev).requires_grad_(True)
    y = R.gn_rows(xr, lv, gm, bt, 32, 1e-5, relu)
    y.backward(torch.cat([_rows(go) for go in gos]).to(torch.bfloat16).to(dev))
    for l, (h, w) in enumerate(sizes):
        sl = slice(lv.row0[l], lv.row0[l] + b * h * w)
        torch.testing.assert_close(_nchw(y[sl].detach().cpu(), b, h, w), refs[l], rtol=2 ** -7, atol=2e-2)
        torch.testing.assert_close(_nchw(xr.grad[sl].cpu(), b, h, w), xs[l].grad, rtol=2 ** -6, atol=3e-2)
    assert _rel(gm.grad.cpu(), gamma.grad) < 5e-3 and _rel(bt.grad.cpu(), beta.grad) < 5e-3


@pytest.mark.parametrize("cfg", [
    # cin, cout, k, stride, pad, relu, residual, scale, sizes
    (64, 256, 1, 1, 0, True, 'add', True, [(9, 14)]),
    (64, 64, 3, 1, 1, True, None, True, [(9, 14)]),
    (256, 128, 1, 2, 0, True, None, True, [(9, 14)]),
    (256, 512, 1, 2, 0, False, None, True, [(8, 10)]),
    (256, 256, 3, 2, 1, False, None, False, [(9, 14)]),
    (512, 256, 1, 1, 0, False, 'nearest', False, [(8, 10)]),
    (256, 256, 3, 1, 1, False, None, False, [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)]),
    (256, 8, 3, 1, 1, False, None, False, [(6, 10), (3, 5)]),
])
def test_conv_rows_forward_backward(cfg):
    from sipmask_amd import hip_ops as H, ops_rows as R
    dev = _dev()
    ci, co, k, stride, pad, relu, res, use_scale, sizes = cfg
    g = torch.Generator().manual_seed(ci + co + k + stride)
    b = 2
    lv = H.Levels(b, sizes)
    xs = [_bf(torch.randn(b, ci, h, w, generator=g)).requires_grad_(True) for h, w in sizes]
    w = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).requires_grad_(True)
    bias = torch.randn(co, generator=g).requires_grad_(True)
    scale = (torch.rand(co, generator=g) + 0.5) if use_scale else None
    weff = w if scale is None else w * scale.view(-1, 1, 1, 1)
    out_sizes = [((h + 2 * pad - k) // stride + 1, (ww + 2 * pad - k) // stride + 1) for h, ww in sizes]
    res_t = res_lv = None
    if res == 'add':
        res_t = [_bf(torch.randn(b, co, h, ww, generator=g)).requires_grad_(True) for h, ww in out_sizes]
    elif res == 'nearest':
        csz = [((h + 1) // 2, (ww + 1) // 2) for h, ww in out_sizes]
        res_t = [_bf(torch.randn(b, co, h, ww, generator=g)).requires_grad_(True) for h, ww in csz]
        res_lv = H.Levels(b, csz)
    gos = [_bf(torch.randn(b, co, h, ww, generator=g)) for h, ww in out_sizes]
    refs = []
    for l, x in enumerate(xs):
        # the kernel multiplies bf16 operands: round the folded weight like sm_weight_prep does
        wq = weff.detach().to(torch.bfloat16).float() + (weff - weff.detach())
        y = F.conv2d(x, wq, bias, stride, pad)
        if res == 'add':
            y = y + res_t[l]
        elif res == 'nearest':
            y = y + F.interpolate(res_t[l], size=y.shape[2:], mode='nearest')
        y = F.relu(y) if relu else y
        y.backward(gos[l])
        refs.append(y.detach())
    xr = torch.cat([_rows(x.detach()) for x in xs]).to(torch.bfloat16).to(dev).requires_grad_(True)
    wd, bd = w.detach().to(dev).requires_grad_(True), bias.detach().to(dev).requires_grad_(True)
    rr = None
    if res_t is not None:
        rr = torch.cat([_rows(t.detach()) for t in res_t]).to(torch.bfloat16).to(dev).requires_grad_(True)
    y, olv = R.conv_rows(xr, lv, wd, bd, stride, pad, relu, None if scale is None else scale.to(dev), rr,
                         res or 'add', res_lv)
    assert olv.sizes == out_sizes
    y.backward(torch.cat([_rows(go) for go in gos]).to(torch.bfloat16).to(dev))
    for l, (h, ww) in enumerate(out_sizes):
        sl = slice(olv.row0[l], olv.row0[l] + b * h * ww)
        torch.testing.assert_close(_nchw(y[sl].detach().cpu(), b, h, ww), refs[l], rtol=2 ** -7, atol=1e-2)
    for l, (h, ww) in enumerate(sizes):
        sl = slice(lv.row0[l], lv.row0[l] + b * h * ww)
        assert _rel(_nchw(xr.grad[sl].cpu(), b, h, ww), xs[l].grad) < 1.5e-2, l
    assert _rel(wd.grad.cpu(), w.grad) < 1.5e-2
    assert _rel(bd.grad.cpu(), bias.grad) < 5e-3
    if res_t is not None:
        rl = res_lv if res == 'nearest' else olv
        for l, (h, ww) in enumerate(rl.sizes):
            sl = slice(rl.row0[l], rl.row0[l] + b * h * ww)
            assert _rel(_nchw(rr.grad[sl].cpu(), b, h, ww), res_t[l].grad) < 1e-2


def test_deform_conv_rows_matches_the_nchw_op():
    """deform_conv_rows over a 3-level pyramid == ops.deform_conv (the reference-interface op, itself held to the oracle
    and the reference's CUDA source by tests/test_gpu_kernels.py / test_ref_pins.py) level by level: outputs and all
    three gradients"""
    from sipmask_amd import hip_ops as H, ops_rows as R, ops as P
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    b, c, co, G = 2, 256, 256, 4
    sizes = [(10, 14), (5, 7), (3, 4)]
    lv = H.Levels(b, sizes)
    xs = [_bf(torch.randn(b, c, h, w, generator=g)).to(dev).requires_grad_(True) for h, w in sizes]
    offs = [(torch.randn(b, G * 18, h, w, generator=g) * 1.5).to(dev).requires_grad_(True) for h, w in sizes]
    w1 = (torch.randn(co, c, 3, 3, generator=g) / 48).to(dev).requires_grad_(True)
    gos = [_bf(torch.randn(b, co, h, w, generator=g)).to(dev) for h, w in sizes]
    refs = []
    for x, o, go in zip(xs, offs, gos):
        y = P.deform_conv(x, o, w1, 1, 1, 1, 1, G)
        y.backward(go)
        refs.append(y.detach())
    gw_ref = w1.grad.clone()
    xr = torch.cat([_rows(x.detach()) for x in xs]).to(torch.bfloat16).requires_grad_(True)
    orows = torch.cat([_rows(o.detach()) for o in offs]).contiguous().requires_grad_(True)
    w2 = w1.detach().clone().requires_grad_(True)
    y = R.deform_conv_rows(xr, lv, orows, w2, None, 1, 1, G)
    y.backward(torch.cat([_rows(go) for go in gos]).to(torch.bfloat16))
    for l, (h, w) in enumerate(sizes):
        sl = slice(lv.row0[l], lv.row0[l] + b * h * w)
        torch.testing.assert_close(_nchw(y[sl].detach(), b, h, w), refs[l], rtol=2 ** -7, atol=1e-2)
        assert _rel(_nchw(xr.grad[sl], b, h, w), xs[l].grad) < 1e-2
        assert _rel(_nchw(orows.grad[sl], b, h, w), offs[l].grad) < 1e-3
    assert _rel(w2.grad, gw_ref) < 1e-3


def test_mask_feat_rows_forward_backward():
    from sipmask_amd import hip_ops as H, ops_rows as R
    dev = _dev()
    g = torch.Generator().manual_seed(6)
    b, c = 2, 64
    sizes = [(8, 12), (4, 6), (2, 3), (1, 2)]
    lv = H.Levels(b, sizes)
    xs = [_bf(torch.randn(b, c, h, w, generator=g)).requires_grad_(True) for h, w in sizes]
    fm = torch.cat([xs[0], F.interpolate(xs[1], scale_factor=2, mode='bilinear', align_corners=False),
                    F.interpolate(xs[2], scale_factor=4, mode='bilinear', align_corners=False)], 1)
    go = _bf(torch.randn(fm.shape, generator=g))
    fm.backward(go)
    xr = torch.cat([_rows(x.detach()) for x in xs]).to(torch.bfloat16).to(dev).requires_grad_(True)
    y = R.mask_feat_rows(xr, lv)
    y.backward(_rows(go).to(torch.bfloat16).to(dev))
    torch.testing.assert_close(_nchw(y.detach().cpu(), b, 8, 12), fm.detach(), rtol=2 ** -7, atol=1e-2)
    for l, (h, w) in enumerate(sizes):
        sl = slice(lv.row0[l], lv.row0[l] + b * h * w)
        got = _nchw(xr.grad[sl].cpu(), b, h, w)
        if l < 3:
            torch.testing.assert_close(got, xs[l].grad, rtol=2 ** -7, atol=2e-2)
        else:
            assert float(got.abs().max()) == 0.0


def test_weight_prep_cache_refreshes_all_stale_operands_in_one_launch():
    """hip_ops.WEIGHT_PREP_CACHE: parameters get persistent operand buffers; refresh() (ops_rows.begin_step) rewrites
    exactly the ones whose parameter / scale changed (in-place update under no_grad = what an optimizer step or a
    checkpoint load does) through sm_weight_prep_multi, with results identical to the single-operand launch"""
    from sipmask_amd import hip_ops as H
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    cache = H._WeightPrepCache()
    shapes = ((64, 3, 7, 8), (256, 64, 1, 64), (72, 128, 3, 128), (8, 256, 3, 256), (128, 512, 1, 512))
    params = [torch.nn.Parameter(torch.randn(co, ci, k, k, generator=g).to(dev)) for co, ci, k, _ in shapes]
    scales = [None, (torch.rand(256, generator=g) + 0.5).to(dev), None, None, (torch.rand(128, generator=g) + 0.5).to(dev)]
    req = [(p, s, m, cp if m == 0 else None) for p, s, (_, _, _, cp) in zip(params, scales, shapes) for m in (0, 1, 2)
           if not (m != 0 and p.shape[1] == 3)]
    first = [cache.get(*r)[0] for r in req]
    assert all(cache.get(*r)[0] is f for r, f in zip(req, first))          # served from the cache, no new buffer
    assert cache.refresh() == 0
    with torch.no_grad():
        for p in params[1:]:
            p.mul_(1.5).add_(0.01)
        scales[1].mul_(0.5)
    n = cache.refresh()
    assert n == len([r for r in req if r[0] is not params[0]])
    for (p, s, m, cp), buf in zip(req, first):
        ref, _ = H.prep_conv_weight(*{0: (p.detach() if s is None else p.detach() * s.view(-1, 1, 1, 1), cp),
                                      1: ((p.detach() if s is None else p.detach() * s.view(-1, 1, 1, 1)).flip(2, 3).permute(1, 0, 2, 3).contiguous(), p.shape[0]),
                                      2: ((p.detach() if s is None else p.detach() * s.view(-1, 1, 1, 1)).permute(2, 3, 1, 0).reshape(-1, p.shape[0], 1, 1).contiguous(), p.shape[0])}[m])
        assert torch.equal(cache.get(p, s, m, cp)[0], ref), (tuple(p.shape), m)
        assert cache.get(p, s, m, cp)[0] is buf
    assert cache.refresh() == 0


@pytest.mark.parametrize("cfg", [
    # cin, cout, k, stride, pad, sizes
    (256, 256, 3, 1, 1, [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)]),      # tower-like pyramid
    (512, 128, 1, 2, 0, [(9, 14)]),                                        # strided 1x1 (layer2.0.conv1)
    (64, 72, 3, 1, 1, [(7, 9)]),                                           # channel tails inside the 128 tiles
    (16, 32, 3, 2, 0, [(21, 17)]),                                         # the SipMask++ scoring convs
    (256, 8, 3, 1, 1, [(6, 10), (3, 5)]),
    (128, 128, 3, 1, 1, [(40, 56)]),                                       # several split-K slices
])
def test_wgrad_direct_vs_gemm_path_and_torch(cfg):
    """sm_wgrad_direct (transposing LDS reads on the NHWC rows) against the materialised im2col^T GEMM path it replaces
    (same bf16 operands, f32 accumulation: only the summation order differs) and against torch's conv2d_weight in f64"""
    from sipmask_amd import hip_ops as H, _lib
    import ctypes as C
    dev = _dev()
    ci, co, k, stride, pad, sizes = cfg
    g = torch.Generator().manual_seed(ci + co + k)
    b = 2
    lv = H.Levels(b, sizes)
    osz = [((h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1) for h, w in sizes]
    olv = H.Levels(b, osz)
    xs = [_bf(torch.randn(b, ci, h, w, generator=g)) for h, w in sizes]
    gos = [_bf(torch.randn(b, co, h, w, generator=g)) for h, w in osz]
    x = torch.cat([_rows(t) for t in xs]).to(torch.bfloat16).to(dev)
    go = torch.cat([_rows(t) for t in gos]).to(torch.bfloat16).to(dev)
    d = H.make_conv_desc(b, sizes, osz, lv.row0, olv.row0, ci, co, co, k, stride, pad, ci, co)
    lib = _lib.load()
    assert lib.sm_wgrad_direct_supported(C.byref(d)) == 1
    gw_new = torch.full((k * k * ci, co), float("nan"), device=dev)
    _lib.check(lib.sm_wgrad_direct(C.byref(d), _lib.ptr(x), _lib.ptr(go), _lib.ptr(gw_new), _lib.stream_ptr()), "sm_wgrad_direct")
    d.flags = 128                                                    # SM_CONV_BWD_WGRAD_GEMM: the round-1 path
    gw_old = torch.empty(k * k * ci, co, device=dev)
    H.conv2d_bwd(d, x, None, None, go, None, gw_old, None)
    ref = sum(torch.nn.grad.conv2d_weight(xi.double(), (co, ci, k, k), gi.double(), stride, pad) for xi, gi in zip(xs, gos))
    ref_t = ref.permute(2, 3, 1, 0).reshape(k * k * ci, co).float()
    scale = float(ref_t.abs().max())
    assert float((gw_new.cpu() - ref_t).abs().max()) <= 2e-5 * scale * max(1.0, (b * sum(h * w for h, w in osz)) ** 0.5 / 8)
    assert float((gw_new - gw_old).abs().max()) <= 2e-5 * scale * max(1.0, (b * sum(h * w for h, w in osz)) ** 0.5 / 8)


def test_offset_linear_rows_vs_torch_autograd():
    """FeatureAlign.conv_offset on the plan's own kernel + its deterministic adjoint (sm_offset_linear_bwd) against the
    torch matmul it replaces in the training graph (sipmask_head.py:30-33,50): forward 1e-6, weight gradient 1e-5 relative,
    two backward passes bit-identical (no float atomics), no gradient into the (detached) box prediction."""
    from sipmask_amd import hip_ops as H
    from sipmask_amd import ops_rows as R
    lv = H.Levels(2, [(25, 31), (13, 16), (7, 8)])
    g = torch.Generator().manual_seed(3)
    box = (torch.rand(lv.rows, 4, generator=g) * 6).cuda()
    w = (torch.randn(72, 4, generator=g) * 0.2).cuda().requires_grad_(True)
    up = torch.randn(lv.rows, 72, generator=g).cuda()
    out = R.offset_linear_rows(box, w, lv)
    ref = box @ w.detach().t()
    assert float((out - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    out.backward(up)
    g1 = w.grad.clone()
    w.grad = None
    R.offset_linear_rows(box, w, lv).backward(up)
    assert torch.equal(g1, w.grad)
    gref = (up.double().t() @ box.double()).float()
    assert float((g1 - gref).norm() / gref.norm()) < 1e-5
