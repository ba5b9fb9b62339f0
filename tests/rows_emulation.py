"""Test infrastructure: a torch-CPU EMULATION of the row-tensor training ops (sipmask_amd/ops_rows.py) with the SAME rounding
points as the HIP kernels, plugged into the SAME Python graph (modules.ResNet.forward_rows / FPN.forward_rows).

Why: the composed backward of the untrained, gain-calibrated trunk is chaotic in its ReLU gates -- two pipelines whose
forward agrees to 0.8 % differ by 30 % in a deep-trunk weight gradient -- so against the fp32 oracle a whole-detector
gradient can only be a wiring check (cosine > 0.85).  What CAN be held tightly is the composition itself: with every op
replaced by plain torch arithmetic that rounds where the kernels round (bf16 operands, f32 accumulation, bf16 storage of
activations and of back-propagated gradients, BatchNorm scale folded into the bf16 weight operand and undone on dW), the
only remaining differences are accumulation orders, and every trunk tensor must agree to ~1e-3.

Rounding points mirrored (ops_rows.ConvRowsFunction, csrc/train_rows.hip):
  forward   y = bf16( act( conv_f32(x_bf16, bf16(w * scale)) + bias + residual_bf16 ) )      (residual: same rows | nearest)
  backward  g = bf16(gout) gated by y > 0;  dX = bf16( conv_transpose_f32(g, bf16(w * scale)) );  dW = f32 sum * scale;
            db = f32 sum;  d(residual) = g  |  bf16( f32 sum of the fine-grid gradients that read a coarse pixel )
"""
import contextlib

import torch
import torch.nn.functional as F
from torch.autograd import Function

BF16 = torch.bfloat16


def _bf(t):
    return t.to(BF16).float()


def _out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def _nchw(rows, b, h, w, c=None):
    t = rows.float().view(b, h, w, rows.shape[1])
    if c is not None:
        t = t[..., :c]
    return t.permute(0, 3, 1, 2).contiguous()


def _rows(t):
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])


def _nearest_src(out_n, in_n):
    """source index of F.interpolate(mode='nearest') / SM_CONV_RES_NEAREST: min(floor(dst * in / out), in - 1), in float32"""
    f = torch.tensor(float(in_n), dtype=torch.float32) / torch.tensor(float(out_n), dtype=torch.float32)
    return torch.clamp(torch.floor(torch.arange(out_n, dtype=torch.float32) * f).long(), max=in_n - 1)


class EmuConvRows(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, scale, residual, cfg):
        from sipmask_amd import hip_ops as H
        lv, stride, pad, relu, res_mode, res_lv, out_f32 = cfg
        co, ci, k, _ = weight.shape
        cs, b = x.shape[1], lv.batch
        wq = weight.detach().float()
        if scale is not None:
            wq = wq * scale.detach().float().view(-1, 1, 1, 1)
        wq = _bf(wq)
        out_sizes = [(_out(h, k, stride, pad), _out(w, k, stride, pad)) for h, w in lv.sizes]
        olv = H.Levels(b, out_sizes)
        ys = []
        for l, (h, w) in enumerate(lv.sizes):
            xl = _nchw(x[lv.row0[l]:lv.row0[l] + b * h * w], b, h, w, ci)
            yl = F.conv2d(xl, wq, None, stride, pad)
            if bias is not None:
                yl = yl + bias.detach().float().view(1, -1, 1, 1)
            if residual is not None:
                oh, ow = out_sizes[l]
                if res_mode == 'nearest':
                    rh, rw = res_lv.sizes[l]
                    r = _nchw(residual[res_lv.row0[l]:res_lv.row0[l] + b * rh * rw], b, rh, rw)
                    r = r[:, :, _nearest_src(oh, rh)][:, :, :, _nearest_src(ow, rw)]
                else:
                    r = _nchw(residual[olv.row0[l]:olv.row0[l] + b * oh * ow], b, oh, ow)
                yl = yl + r
            if relu:
                yl = torch.relu(yl)
            ys.append(_rows(yl))
        y = torch.cat(ys)
        y = y if out_f32 else y.to(BF16)
        ctx.cfg = (lv, olv, stride, pad, relu, res_mode, res_lv, k, co, ci, cs, bias is not None)
        ctx.save_for_backward(x, wq, scale, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, wq, scale, y = ctx.saved_tensors
        lv, olv, stride, pad, relu, res_mode, res_lv, k, co, ci, cs, has_bias = ctx.cfg
        b = lv.batch
        g = g.to(BF16)
        if relu:
            g = torch.where(y.to(BF16) > 0, g, torch.zeros_like(g))
        gx = gw = gb = g_res = None
        if has_bias and ctx.needs_input_grad[2]:
            gb = g.float().sum(0)
        if ctx.needs_input_grad[4]:
            if res_mode == 'nearest':
                parts = []
                for l, (oh, ow) in enumerate(olv.sizes):
                    rh, rw = res_lv.sizes[l]
                    gl = _nchw(g[olv.row0[l]:olv.row0[l] + b * oh * ow], b, oh, ow)
                    acc = torch.zeros(b, co, rh, ow)
                    acc.index_add_(2, _nearest_src(oh, rh), gl)
                    acc2 = torch.zeros(b, co, rh, rw)
                    acc2.index_add_(3, _nearest_src(ow, rw), acc)
                    parts.append(_rows(acc2).to(BF16))
                g_res = torch.cat(parts)
            else:
                g_res = g
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gxs, gw = [], torch.zeros_like(wq)
            for l, (h, w) in enumerate(lv.sizes):
                oh, ow = olv.sizes[l]
                xl = _nchw(x[lv.row0[l]:lv.row0[l] + b * h * w], b, h, w, ci)
                gl = _nchw(g[olv.row0[l]:olv.row0[l] + b * oh * ow], b, oh, ow)
                if ctx.needs_input_grad[0]:
                    assert ci == cs
                    gxs.append(_rows(torch.nn.grad.conv2d_input(xl.shape, wq, gl, stride, pad)).to(BF16))
                if ctx.needs_input_grad[1]:
                    gw = gw + torch.nn.grad.conv2d_weight(xl, wq.shape, gl, stride, pad)
            if ctx.needs_input_grad[0]:
                gx = torch.cat(gxs)
            if ctx.needs_input_grad[1]:
                if scale is not None:
                    gw = gw * scale.detach().float().view(-1, 1, 1, 1)
            else:
                gw = None
        return gx, gw, gb, None, g_res, None


def _emu_nchw_to_rows(x, cpad=None):
    b, c, h, w = x.shape
    cpad = cpad or (c + 7) // 8 * 8
    y = torch.zeros(b * h * w, cpad, dtype=BF16)
    y[:, :c] = _rows(x.detach().float()).to(BF16)
    return y


def _emu_maxpool(x, y, b, h, w, c):
    y.copy_(_rows(F.max_pool2d(_nchw(x, b, h, w), 3, 2, 1)).to(BF16))
    return y


@contextlib.contextmanager
def emulated_rows():
    """inside: ops_rows.conv_rows / nchw_to_rows and hip_ops.maxpool3x3s2 run as the CPU emulation (CPU tensors)"""
    from sipmask_amd import hip_ops as H
    from sipmask_amd import ops_rows as R
    saved = (R.ConvRowsFunction, R.nchw_to_rows, H.maxpool3x3s2)
    R.ConvRowsFunction, R.nchw_to_rows, H.maxpool3x3s2 = EmuConvRows, _emu_nchw_to_rows, _emu_maxpool
    try:
        yield
    finally:
        R.ConvRowsFunction, R.nchw_to_rows, H.maxpool3x3s2 = saved


def rows_to_nchw(rows, b, h, w):
    return rows.float().view(b, h, w, rows.shape[1]).permute(0, 3, 1, 2)
