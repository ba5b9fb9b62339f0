"""Autograd ops of the training graph on NHWC bf16 ROW tensors (BASELINE config #4).

The reference trains through ATen/cuDNN layer by layer on NCHW float tensors (M/mmdet/models/backbones/resnet.py:205-239,
necks/fpn.py:137-178, anchor_heads/sipmask_head.py:241-287).  The first version here (ops.py: conv2d / group_norm /
deform_conv on NCHW float tensors) kept that interface per op and paid for it between the ops: 19 ms of a 62 ms step
were ATen copies, casts, fills and adds, 7 ms NCHW<->NHWC transposes (profiles/r02h_train_kernel_stats_before.csv).
These ops exchange what the MFMA kernels read and write -- `[positions, channels]` bf16 matrices, all images of all
pyramid levels in one tensor (hip_ops.Levels carries the geometry) -- and fuse bias / frozen-BN fold / ReLU / residual
into the conv launches:

  conv_rows         conv (+ per-cout scale folded into the weight at prep time, + bias, + residual add or nearest-upsampled
                    residual, + ReLU), multi-level; backward = ReLU gate, dX as a conv with the flipped weights (bf16 rows
                    out), dW through sm_conv2d_bwd, both operand layouts produced by one sm_weight_prep launch each
  gn_rows           GroupNorm (+ReLU) over pyramid rows, statistics per (image, level, group)
  deform_conv_rows  FeatureAlign's deformable conv over the whole pyramid
  mask_feat_rows    the [l0 | up2(l1) | up4(l2)] concatenation the mask branch reads (sipmask_head.py:266-275)

Gradients w.r.t. activations are bf16 rows, w.r.t. parameters f32 in the parameter's own layout.  CPU tensors raise
NotImplementedError (no fallback).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from . import hip_ops as H

BF16 = torch.bfloat16
SM_CONV_BWD_GX_BF16 = 64


def _out_size(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


def _need_cuda(t):
    if not t.is_cuda:
        raise NotImplementedError("sipmask_amd row ops are HIP-only")


def _big_tile_flags(k, stride, cout, rows):
    """The 256 x 256 8-wave tile with the hand-placed K step (include/sipmask_hip.h: plan selectors, identical results) for the
    wide 3x3 convs over the whole pyramid -- the towers and FeatureAlign's neighbours, forward and dX: 0.109 instead of 0.121
    ms per conv at B=4 (the library's own rule keeps the 128 x 128 tile unless asked; the inference engine asks too)."""
    if k == 3 and stride == 1 and cout % 256 == 0 and rows >= 60000:
        return _lib.SM_CONV_DBG_TILE256 | _lib.SM_CONV_DBG_HAND_PLACED
    return 0


class ConvRowsFunction(Function):
    """y = act(conv(x; weight * scale[:, None, None, None]) + bias [+ residual]) on row tensors.

    x bf16 [rows_in, cin_stride]; weight f32 [co, ci, k, k] (ci <= cin_stride: the stem's 3 channels live in 8);
    scale f32 [co] or None (constant: the eval-mode BatchNorm fold); bias f32 [co] or None; residual bf16
    [rows_out or res rows, co] or None.  cfg = (lv_in, stride, pad, relu, res_mode, res_lv, out_f32):
    res_mode 'add' (same rows) | 'nearest' (FPN top-down: residual lives on the coarser grid res_lv)."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale, residual, cfg):
        _need_cuda(x)
        lv, stride, pad, relu, res_mode, res_lv, out_f32 = cfg
        co, ci, k, k2 = weight.shape
        cs = x.shape[1]
        if k != k2 or ci > cs or cs % 8 != 0 or co % 8 != 0 or x.dtype != BF16 or not x.is_contiguous():
            raise NotImplementedError("conv_rows: square kernels, bf16 contiguous rows, channels a multiple of 8")
        if x.shape[0] != lv.rows:
            raise ValueError("conv_rows: %d rows for a pyramid of %d" % (x.shape[0], lv.rows))
        out_sizes = [(_out_size(h, k, stride, pad), _out_size(w, k, stride, pad)) for h, w in lv.sizes]
        olv = H.Levels(lv.batch, out_sizes)
        wq, co_pad = H.weight_prep(weight, scale, 0, cs)
        flags = (_lib.SM_CONV_RELU if relu else 0) | (_lib.SM_CONV_OUT_F32 if out_f32 else 0)
        flags |= _big_tile_flags(k, stride, co, olv.rows)
        res_cs, res_sizes, res_row0 = 0, None, None
        if residual is not None:
            if residual.dtype != BF16 or not residual.is_contiguous() or residual.shape[1] != co:
                raise NotImplementedError("conv_rows: bf16 contiguous residual with cout channels")
            res_cs = co
            if res_mode == 'nearest':
                flags |= _lib.SM_CONV_RES_NEAREST
                res_sizes, res_row0 = res_lv.sizes, res_lv.row0
            else:
                flags |= _lib.SM_CONV_RES_ADD
        d = H.make_conv_desc(lv.batch, lv.sizes, out_sizes, lv.row0, olv.row0, cs, co, co_pad, k, stride, pad, cs, co,
                             flags=flags, res_cstride=res_cs, res_sizes=res_sizes, res_row0=res_row0)
        y = torch.empty(olv.rows, co, dtype=torch.float32 if out_f32 else BF16, device=x.device)
        b = None if bias is None else bias.detach().float().contiguous()
        H.conv2d(d, x, wq, b, residual, y)
        ctx.cfg = (lv, olv, stride, pad, relu, res_mode, res_lv, out_f32, k, co, ci, cs, bias is not None)
        ctx.bias_ref = bias.detach() if (bias is not None and bias.dtype == torch.float32) else None
        ctx.save_for_backward(x, weight, scale, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, weight, scale, y = ctx.saved_tensors
        lv, olv, stride, pad, relu, res_mode, res_lv, out_f32, k, co, ci, cs, has_bias = ctx.cfg
        need_x, need_w, need_b, need_res = (ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                            has_bias and ctx.needs_input_grad[2], ctx.needs_input_grad[4])
        g = g.contiguous()
        if g.dtype != BF16:
            g = g.to(BF16)
        if relu:
            g = H.relu_bwd_bf16(g, y if y.dtype == BF16 else y.to(BF16))
        g_res = None
        if need_res:
            if res_mode == 'nearest':
                parts = [H.nearest_bwd_rows(g[olv.row0[l]:olv.row0[l] + olv.batch * h * w], olv.batch, (h, w), res_lv.sizes[l], co)
                         for l, (h, w) in enumerate(olv.sizes)]
                g_res = parts[0] if len(parts) == 1 else torch.cat(parts)
            else:
                g_res = g
        gb = None
        if need_b:
            bias_p = ctx.bias_ref
            sink = H.GRAD_SINK.target(bias_p) if co <= 256 else None
            gb = H.bias_grad_rows(g, co, out=sink) if co <= 256 else g.float().sum(0)
            if sink is not None:          # written straight into the all-reduce bucket (hip_ops._GradSink)
                H.GRAD_SINK.commit(bias_p)
                gb = None
        gx = gw = None
        if need_x or need_w:
            if need_x and ci != cs:
                raise NotImplementedError("conv_rows: input gradient needs cin == the row stride")
            d = H.make_conv_desc(lv.batch, lv.sizes, olv.sizes, lv.row0, olv.row0, cs, co, co, k, stride, pad, cs, co)
            w_dg = gx_buf = None
            if need_x and stride == 1:
                w_dg, _ = H.weight_prep(weight, scale, 1)
                d.flags = SM_CONV_BWD_GX_BF16 | _big_tile_flags(k, stride, cs, lv.rows)
                gx_buf = torch.empty(lv.rows, cs, dtype=BF16, device=g.device)
            gw_t = torch.empty(k * k * cs, co, dtype=torch.float32, device=g.device) if need_w else None
            if gx_buf is not None or gw_t is not None:
                H.conv2d_bwd(d, x, None, w_dg, g, gx_buf, gw_t, None)
            if need_x and stride == 1:
                gx = gx_buf
            elif need_x:
                # strided conv: dX = (stride-1 conv with the flipped weights) of the gradient placed on the stride-1 output
                # grid -- for a 1x1 conv that is a channel GEMM on the strided grid followed by the scatter, for k > 1
                # the scatter (zero-dilation) comes first and the conv pads by k-1-pad.  Both stay on the bf16 MFMA path
                # for any channel count (the grad-column + col2im route of sm_conv2d_bwd needs cin % 64 == 0 and f32 atomics).
                w_dg, rp = H.weight_prep(weight, scale, 1)
                parts = []
                for l, (oh, ow) in enumerate(olv.sizes):
                    h, w = lv.sizes[l]
                    gl = g[olv.row0[l]:olv.row0[l] + olv.batch * oh * ow]
                    if k == 1:
                        d1 = H.make_conv_desc(olv.batch, [(oh, ow)], [(oh, ow)], [0], [0], co, cs, rp, 1, 1, 0, co, cs)
                        t = torch.empty(olv.batch * oh * ow, cs, dtype=BF16, device=g.device)
                        H.conv2d(d1, gl, w_dg, None, None, t)
                        parts.append(H.scatter_stride_rows(t, lv.batch, (h, w), (oh, ow), stride, cs))
                    else:
                        hd, wd = h + 2 * pad - k + 1, w + 2 * pad - k + 1          # stride-1 output grid
                        dil = H.scatter_stride_rows(gl, lv.batch, (hd, wd), (oh, ow), stride, co)
                        d1 = H.make_conv_desc(lv.batch, [(hd, wd)], [(h, w)], [0], [0], co, cs, rp, k, 1, k - 1 - pad, co, cs)
                        t = torch.empty(lv.batch * h * w, cs, dtype=BF16, device=g.device)
                        H.conv2d(d1, dil, w_dg, None, None, t)
                        parts.append(t)
                gx = parts[0] if len(parts) == 1 else torch.cat(parts)
            if need_w:
                sink = H.GRAD_SINK.target(weight) if (cs == ci and weight.dtype == torch.float32) else None
                gw = H.wgrad_finish(gw_t, scale, co, cs, k, k, out=sink)
                if sink is not None:
                    H.GRAD_SINK.commit(weight)
                    gw = None
                elif cs != ci:
                    gw = gw[:, :ci].contiguous()
        return gx, gw, gb, None, g_res, None


def conv_rows(x, lv, weight, bias=None, stride=1, pad=0, relu=False, scale=None, residual=None, res_mode='add',
              res_lv=None, out_f32=False):
    """-> (y rows, Levels of y)"""
    k = weight.shape[2]
    olv = H.Levels(lv.batch, [(_out_size(h, k, stride, pad), _out_size(w, k, stride, pad)) for h, w in lv.sizes])
    y = ConvRowsFunction.apply(x, weight, bias, scale, residual, (lv, stride, pad, relu, res_mode, res_lv, out_f32))
    return y, olv


class GroupNormRowsFunction(Function):
    """F.group_norm(+ReLU) of pyramid rows: statistics per (image, level, group) (sm_groupnorm / sm_gn_bwd_rows)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, lv, groups, eps, relu):
        _need_cuda(x)
        c = x.shape[1]
        if x.dtype != BF16 or not x.is_contiguous() or x.shape[0] != lv.rows:
            raise NotImplementedError("gn_rows: bf16 contiguous pyramid rows")
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        stats = H.gn_stats_alloc(lv.batch * len(lv) * groups, x.device)
        y = torch.empty_like(x)
        H.groupnorm(x, y, g32, b32, stats, lv, c, groups, eps, relu)
        ctx.save_for_backward(x, g32, b32, stats)
        ctx.cfg = (lv, groups, eps, relu, c)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, g32, b32, stats = ctx.saved_tensors
        lv, groups, eps, relu, c = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        # g32 / b32 share the parameters' storage (f32 contiguous parameters), so they address the gradient sink
        sg, sb = H.GRAD_SINK.target(g32), H.GRAD_SINK.target(b32)
        if sg is None or sb is None:
            sg = sb = None
        dx, dg, db = H.gn_bwd_rows(x, dy, g32, b32, stats, lv, c, groups, eps, relu, dg=sg, db=sb)
        if sg is not None:
            H.GRAD_SINK.commit(g32)
            H.GRAD_SINK.commit(b32)
            dg = db = None
        return dx, dg, db, None, None, None, None


def gn_rows(x, lv, gamma, beta, groups=32, eps=1e-5, relu=True):
    return GroupNormRowsFunction.apply(x, gamma, beta, lv, groups, eps, relu)


class DeformConvRowsFunction(Function):
    """Deformable conv v1 (3x3-style, stride 1) over pyramid rows: x bf16 [rows, c], offset f32 [rows, G*k*k*2] in the
    reference channel order (deform_conv_cuda_kernel.cu:216-223), weight f32 OIHW (x scale), optional bias / ReLU."""

    @staticmethod
    def forward(ctx, x, offset, weight, bias, scale, cfg):
        _need_cuda(x)
        lv, pad, dil, g, relu = cfg
        co, ci, k, _ = weight.shape
        if x.dtype != BF16 or not x.is_contiguous() or x.shape[1] != ci or ci % (8 * g) != 0 or co % 8 != 0:
            raise NotImplementedError("deform_conv_rows: bf16 contiguous rows, channels a multiple of 8*deformable_groups")
        off = offset.detach().float().contiguous()
        if off.shape != (lv.rows, g * 2 * k * k):
            raise ValueError("invalid offset shape {} (expected {})".format(tuple(off.shape), (lv.rows, g * 2 * k * k)))
        wq, co_pad = H.weight_prep(weight, scale, 0, ci)
        d = H.make_conv_desc(lv.batch, lv.sizes, lv.sizes, lv.row0, lv.row0, ci, co, co_pad, k, 1, pad, ci, co,
                             flags=_lib.SM_CONV_RELU if relu else 0, dil=dil, deform_groups=g)
        y = torch.empty(lv.rows, co, dtype=BF16, device=x.device)
        H.deform_conv2d(d, x, off, wq, None if bias is None else bias.detach().float().contiguous(), y)
        ctx.cfg = (lv, pad, dil, g, relu, k, co, ci, bias is not None)
        ctx.save_for_backward(x, off, weight, scale, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, off, weight, scale, y = ctx.saved_tensors
        lv, pad, dil, g, relu, k, co, ci, has_bias = ctx.cfg
        if ci % 64 != 0 or (ci // g) % 64 != 0:
            raise NotImplementedError("deform conv backward needs 64 | channels per deformable group")
        gout = gout.contiguous()
        if gout.dtype != BF16:
            gout = gout.to(BF16)
        if relu:
            gout = H.relu_bwd_bf16(gout, y)
        need_in = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_w = ctx.needs_input_grad[2]
        gb = H.bias_grad_rows(gout, co) if has_bias and ctx.needs_input_grad[3] and co <= 256 else \
            (gout.float().sum(0) if has_bias and ctx.needs_input_grad[3] else None)
        d = H.make_conv_desc(lv.batch, lv.sizes, lv.sizes, lv.row0, lv.row0, ci, co, co, k, 1, pad, ci, co, dil=dil,
                             deform_groups=g)
        w_t = H.weight_prep(weight, scale, 2)[0] if need_in else None
        gx = torch.empty(lv.rows, ci, dtype=torch.float32, device=x.device) if need_in else None
        goff = torch.empty_like(off) if need_in else None
        gw_t = torch.empty(k * k * ci, co, dtype=torch.float32, device=x.device) if need_w else None
        H.deform_conv2d_bwd(d, x, off, w_t, gout, gx, goff, gw_t)
        gw = None
        if need_w:
            sink = H.GRAD_SINK.target(weight) if weight.dtype == torch.float32 else None
            gw = H.wgrad_finish(gw_t, scale, co, ci, k, k, out=sink)
            if sink is not None:
                H.GRAD_SINK.commit(weight)
                gw = None
        return (None if gx is None else gx.to(BF16)), goff, gw, gb, None, None


def deform_conv_rows(x, lv, offset, weight, bias=None, pad=1, dil=1, deformable_groups=1, relu=False, scale=None):
    return DeformConvRowsFunction.apply(x, offset, weight, bias, scale, (lv, pad, dil, deformable_groups, relu))


class OffsetLinearRowsFunction(Function):
    """FeatureAlign.conv_offset on rows (sipmask_head.py:30-33,50): offset [rows, nout] = box [rows, 4] . W^T, a 1x1 conv
    without bias of the DETACHED box prediction -- only the weight receives a gradient.  Forward = the inference plan's
    sm_offset_linear, backward = sm_offset_linear_bwd (deterministic two-pass reduction); the torch matmul this replaces
    put a vendor GEMM (K = every position of the pyramid) on the training path."""

    @staticmethod
    def forward(ctx, box, w_off, lv):
        _need_cuda(box)
        box = box.detach()
        if box.dtype != torch.float32 or not box.is_contiguous() or box.shape[1] % 4 != 0 or box.shape[0] != lv.rows:
            raise NotImplementedError("offset_linear_rows: contiguous f32 rows [lv.rows, 4k]")
        w = w_off.detach().float().contiguous()
        out = torch.empty(lv.rows, w.shape[0], dtype=torch.float32, device=box.device)
        H.offset_linear(box, box.shape[1], w, lv, out)
        ctx.save_for_backward(box)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (box,) = ctx.saved_tensors
        gw = H.offset_linear_bwd(box, box.shape[1], gout.float().contiguous()) if ctx.needs_input_grad[1] else None
        return None, gw, None


def offset_linear_rows(box, w_off, lv):
    """box f32 [rows, 4] (no gradient flows into it), w_off f32 [nout, 4] -> offsets f32 [rows, nout]"""
    return OffsetLinearRowsFunction.apply(box, w_off, lv)


class MaskFeatRowsFunction(Function):
    """[level 0 | bilinear x2 of level 1 | bilinear x4 of level 2] of a pyramid row tensor as one [B*H0*W0, 3c] matrix
    (sipmask_head.py:266-275: the feat_masks list the mask branch concatenates); the upsampling kernel writes straight
    into its channel slice, the backward gathers from it."""

    @staticmethod
    def forward(ctx, x, lv):
        _need_cuda(x)
        c = x.shape[1]
        b = lv.batch
        h0, w0 = lv.sizes[0]
        for l in (1, 2):
            if (lv.sizes[l][0] * 2 ** l, lv.sizes[l][1] * 2 ** l) != (h0, w0):
                raise NotImplementedError("mask_feat_rows: levels 1 / 2 must be exactly 1/2 and 1/4 of level 0")
        out = torch.empty(b * h0 * w0, 3 * c, dtype=BF16, device=x.device)
        out[:, :c] = x[:b * h0 * w0]
        for l in (1, 2):
            h, w = lv.sizes[l]
            H.upsample_bilinear(x[lv.row0[l]:lv.row0[l] + b * h * w], out, b, h, w, c, 2 ** l, c, 3 * c, l * c, False)
        ctx.cfg = (lv, c)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        lv, c = ctx.cfg
        g = g.contiguous()
        if g.dtype != BF16:
            g = g.to(BF16)
        b = lv.batch
        gx = torch.zeros(lv.rows, c, dtype=BF16, device=g.device)
        h0, w0 = lv.sizes[0]
        gx[:b * h0 * w0] = g[:, :c]
        for l in (1, 2):
            h, w = lv.sizes[l]
            gx[lv.row0[l]:lv.row0[l] + b * h * w] = H.upsample_bilinear_bwd_rows(g, 3 * c, l * c, b, h, w, c, 2 ** l)
        return gx, None


def mask_feat_rows(x, lv):
    return MaskFeatRowsFunction.apply(x, lv)


# ------------------------------------------------------------------------------- layout boundary
def nchw_to_rows(x, cpad=None):
    """NCHW float tensor -> bf16 rows [B*H*W, cpad] (no gradient: used for images and caller-provided features)"""
    b, c, h, w = x.shape
    cpad = cpad or (c + 7) // 8 * 8
    y = torch.empty(b * h * w, cpad, dtype=BF16, device=x.device)
    H.nchw_to_nhwc_bf16(x.detach().float().contiguous(), y, cpad)
    return y


class RowsFromNCHW(Function):
    """differentiable NCHW float -> bf16 rows (the head's entry when a caller hands it NCHW features)"""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        ctx.dt = x.dtype
        return nchw_to_rows(x, x.shape[1])

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        b, c, h, w = ctx.shape
        return g.view(b, h, w, c).permute(0, 3, 1, 2).to(ctx.dt)


def rows_to_nchw(y, batch, h, w):
    """rows [B*H*W, C] -> an NCHW VIEW (channels_last strides); differentiable, no copy"""
    return y.view(batch, h, w, y.shape[1]).permute(0, 3, 1, 2)


def begin_step():
    """start of a training step: re-lay out every cached conv-parameter operand that changed since the last step in one
    launch (hip_ops.WEIGHT_PREP_CACHE).  Optional -- without it each operand is refreshed by its own launch on first use."""
    return H.WEIGHT_PREP_CACHE.refresh()


_FOLD_CACHE = {}


def bn_fold_constants(bn):
    """(scale, shift) f32 [C] of an eval-mode BatchNorm whose affine parameters are frozen: y = x*scale + shift.
    Cached on the versions of the four tensors (a checkpoint load or an in-place edit recomputes)."""
    key = id(bn)
    ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.running_var.data_ptr(), bn.weight.device)
    hit = _FOLD_CACHE.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1], hit[2]
    with torch.no_grad():
        s = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
        b = (bn.bias - bn.running_mean * s).float().contiguous()
    _FOLD_CACHE[key] = (ver, s, b)
    return s, b
