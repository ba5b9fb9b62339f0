"""Host-side mirror of the reference's ``mmdet.ops`` / ``mmdet.core.post_processing`` entry points
for the SipMask hot path -- same names, argument meaning and error behaviour, HIP underneath.

  DeformConv / deform_conv      M/mmdet/ops/dcn/deform_conv.py:16-96,189-255
  CropSplit / crop_split        M/mmdet/ops/crop/crop_split.py:9-50
  CropSplitGt / crop_split_gt   M/mmdet/ops/crop/crop_split_gt.py:9-38
  nms                           M/mmdet/ops/nms/nms_wrapper.py:7-60
  sigmoid_focal_loss            M/mmdet/ops/sigmoid_focal_loss/sigmoid_focal_loss.py:10-43
  multiclass_nms_idx            M/mmdet/core/post_processing/bbox_nms.py:79-146
  Scale                         M/mmdet/ops/scale.py:5-15

There is no CPU implementation (as in the reference): CPU tensors raise
NotImplementedError, a missing libsipmask_hip.so raises RuntimeError.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair, _single

from . import _lib
from . import hip_ops as H


def _rows_bf16(t, c):
    """NCHW float tensor -> bf16 NHWC rows [B*H*W, c].  A channels_last-strided tensor (what the conv ops below
    return: an NHWC buffer viewed as NCHW) is just cast -- no transposition; an NCHW-contiguous one goes through
    the transposing kernel."""
    b, _, h, w = t.shape
    t = t.detach()
    nhwc = t.permute(0, 2, 3, 1)
    if nhwc.is_contiguous():
        return nhwc.reshape(b * h * w, c).to(torch.bfloat16)
    x = torch.empty(b * h * w, c, dtype=torch.bfloat16, device=t.device)
    H.nchw_to_nhwc_bf16(t.float().contiguous(), x, c)
    return x


# ------------------------------------------------------------------------------- deform conv
class DeformConvFunction(Function):
    """One conv group (groups == 1) of M/mmdet/ops/dcn/deform_conv.py:16-96 on the C ABI: kh x kw kernels, one stride /
    dilation for both axes (the descriptor's), symmetric padding (deform_conv() pads explicitly when the axes differ)."""

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        if not input.is_cuda:
            raise NotImplementedError
        stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
        if groups != 1:
            raise NotImplementedError("DeformConvFunction is one conv group; deform_conv() splits groups > 1")
        if stride[0] != stride[1] or dilation[0] != dilation[1]:
            raise NotImplementedError("sipmask_amd DeformConv: one stride and one dilation for both axes (the C ABI's "
                                      "sm_conv_desc carries one of each)")
        if padding[0] != padding[1]:
            raise NotImplementedError("DeformConvFunction takes symmetric padding; deform_conv() pads per axis")
        b, c, h, w = input.shape
        co, ci, kh, kw = weight.shape
        g = deformable_groups
        ho = (h + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
        wo = (w + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
        if min(ho, wo) <= 0:
            raise ValueError("convolution input is too small (output would be {}x{})".format(ho, wo))
        if offset.shape != (b, g * 2 * kh * kw, ho, wo):
            raise ValueError("invalid offset shape {} (expected {})".format(tuple(offset.shape),
                                                                           (b, g * 2 * kh * kw, ho, wo)))
        if c % 8 != 0 or c % (8 * g) != 0:
            raise NotImplementedError("channels must be a multiple of 8*deformable_groups")
        cur_im2col_step = min(im2col_step, b)
        assert (b % cur_im2col_step) == 0, "im2col step must divide batchsize"
        x = _rows_bf16(input, c)
        off = offset.detach().float().permute(0, 2, 3, 1).contiguous().view(b * ho * wo, -1)
        wq, co_pad = H.prep_conv_weight(weight.detach())
        y = torch.empty(b * ho * wo, co, dtype=torch.float32, device=input.device)
        d = H.make_conv_desc(b, [(h, w)], [(ho, wo)], [0], [0], c, co, co_pad, (kh, kw), stride[0], padding[0], c, co,
                             flags=_lib.SM_CONV_OUT_F32, dil=dilation[0], deform_groups=g)
        H.deform_conv2d(d, x, off, wq, None, y)
        if input.requires_grad or offset.requires_grad or weight.requires_grad:
            ctx.save_for_backward(x, off, weight)
            ctx.geom = (b, c, h, w, co, kh, kw, ho, wo, stride[0], padding[0], dilation[0], g, input.dtype)
        return y.view(b, ho, wo, co).permute(0, 3, 1, 2).to(input.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        """deform_conv.py:60-96: (grad_input, grad_offset, grad_weight) through sm_deform_conv2d_bwd.
        Mixed precision like the forward: bf16 operands (x, weight, grad_output), f32 accumulation."""
        if not grad_output.is_cuda:
            raise NotImplementedError
        x, off, weight = ctx.saved_tensors
        b, c, h, w, co, kh, kw, ho, wo, stride, pad, dil, g, dt = ctx.geom
        dev = grad_output.device
        if c % 64 != 0 or (c // g) % 64 != 0 or co % 8 != 0:
            raise NotImplementedError("deform conv backward needs 64 | channels per deformable group and 8 | out_channels")
        go = _rows_bf16(grad_output, co)
        d = H.make_conv_desc(b, [(h, w)], [(ho, wo)], [0], [0], c, co, co, (kh, kw), stride, pad, c, co,
                             dil=dil, deform_groups=g)
        need_in = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_w = ctx.needs_input_grad[2]
        K = kh * kw * c
        w_t = None
        if need_in:      # W^T as the operand of the grad-column GEMM: a 1x1 conv weight [K][co]
            w_t, _ = H.prep_conv_weight(weight.detach().permute(2, 3, 1, 0).reshape(K, co, 1, 1).contiguous(), co)
        gx = torch.empty(b * h * w, c, dtype=torch.float32, device=dev) if need_in else None
        goff = torch.empty_like(off) if need_in else None
        gw_t = torch.empty(K, co, dtype=torch.float32, device=dev) if need_w else None
        H.deform_conv2d_bwd(d, x, off, w_t, go, gx, goff, gw_t)
        grad_input = grad_offset = grad_weight = None
        if need_in:
            grad_input = gx.view(b, h, w, c).permute(0, 3, 1, 2).to(dt)
            grad_offset = goff.view(b, ho, wo, -1).permute(0, 3, 1, 2).contiguous()
        if need_w:
            grad_weight = gw_t.view(kh, kw, c, co).permute(3, 2, 0, 1).contiguous().to(weight.dtype)
        return (grad_input, grad_offset, grad_weight, None, None, None, None, None, None)


def deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
    """M/mmdet/ops/dcn/deform_conv.py:99 (`deform_conv = DeformConvFunction.apply`) with the op's full argument range:
      * groups > 1: one launch per conv group over its channel slice (the reference's grouped GEMM,
        deform_conv_cuda.cpp:231-236); the deformable groups follow the CHANNELS (channel c samples with the offsets of
        deformable group c // (C / deformable_groups), deform_conv_cuda_kernel.cu:206), so a conv group takes the
        deformable groups its slice covers -- whole ones when deformable_groups is a multiple of groups, the one it lies in
        when groups is a multiple of deformable_groups; autograd sums the offset gradients of groups that share one;
      * padding that differs between the axes: the input is zero-padded explicitly and the op runs unpadded -- the same
        values, because a bilinear corner outside the image contributes zero either way (kernel.cu:98-109);
      * stride / dilation must be the same for both axes (NotImplementedError otherwise)."""
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    if input is not None and input.dim() != 4:
        raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
    if padding[0] != padding[1]:
        input = torch.nn.functional.pad(input, (padding[1], padding[1], padding[0], padding[0]))
        padding = (0, 0)
    if groups == 1:
        return DeformConvFunction.apply(input, offset, weight, stride, padding, dilation, 1, deformable_groups, im2col_step)
    c, co = input.shape[1], weight.shape[0]
    if c % groups or co % groups or weight.shape[1] * groups != c:
        raise ValueError("channels {} / out_channels {} / weight {} do not fit groups={}".format(
            c, co, tuple(weight.shape), groups))
    dg, kk2 = deformable_groups, 2 * weight.shape[2] * weight.shape[3]
    if dg % groups != 0 and groups % dg != 0:
        raise NotImplementedError("groups={} and deformable_groups={}: one must divide the other".format(groups, dg))
    cg, cog, outs = c // groups, co // groups, []
    for gi in range(groups):
        if dg % groups == 0:
            sub = dg // groups
            off = offset[:, gi * sub * kk2:(gi + 1) * sub * kk2]
        else:
            sub = 1
            di = gi // (groups // dg)
            off = offset[:, di * kk2:(di + 1) * kk2]
        outs.append(DeformConvFunction.apply(input[:, gi * cg:(gi + 1) * cg], off, weight[gi * cog:(gi + 1) * cog], stride,
                                             padding, dilation, 1, sub, im2col_step))
    return torch.cat(outs, 1)




class DeformConv(nn.Module):
    """Same constructor / parameters / forward contract as M/mmdet/ops/dcn/deform_conv.py:189-255."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super(DeformConv, self).__init__()
        # M/ asserts `not bias` (deform_conv.py:203); the maskrcnn-benchmark module takes one
        # (B/fcos_core/layers/dcn/deform_conv_module.py:10-50), which its FeatureAlign uses
        self.with_bias = bool(bias)
        assert in_channels % groups == 0, \
            'in_channels {} cannot be divisible by groups {}'.format(in_channels, groups)
        assert out_channels % groups == 0, \
            'out_channels {} cannot be divisible by groups {}'.format(out_channels, groups)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.transposed = False
        self.output_padding = _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // self.groups, *self.kernel_size))
        if self.with_bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        # input smaller than the kernel: pad, run, crop (deform_conv.py:239-255)
        input_pad = (x.size(2) < self.kernel_size[0] or x.size(3) < self.kernel_size[1])
        if input_pad:
            pad_h = max(self.kernel_size[0] - x.size(2), 0)
            pad_w = max(self.kernel_size[1] - x.size(3), 0)
            x = torch.nn.functional.pad(x, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
            offset = torch.nn.functional.pad(offset, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
        out = deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                          self.deformable_groups)
        if input_pad:
            out = out[:, :, :out.size(2) - pad_h, :out.size(3) - pad_w].contiguous()
        if self.with_bias:
            out = out + self.bias.view(1, -1, 1, 1)
        return out


# ------------------------------------------------------------------------------- plain conv (fwd + bwd)
class Conv2dFunction(Function):
    """F.conv2d(input, weight, bias, stride, padding, dilation) on the HIP implicit-GEMM kernels, forward and
    backward (bf16 operands, f32 accumulation) -- the building block of the training step.
    NCHW float tensors in and out, square kernels, groups=1."""

    @staticmethod
    def forward(ctx, input, weight, bias=None, stride=1, padding=0, dilation=1):
        if not input.is_cuda:
            raise NotImplementedError
        b, c, h, w = input.shape
        co, ci, k, k2 = weight.shape
        if k != k2 or ci != c or c % 8 != 0 or co % 8 != 0:
            raise NotImplementedError("square kernels, channels a multiple of 8")
        ho = (h + 2 * padding - (dilation * (k - 1) + 1)) // stride + 1
        wo = (w + 2 * padding - (dilation * (k - 1) + 1)) // stride + 1
        x = _rows_bf16(input, c)
        wq, co_pad = H.prep_conv_weight(weight.detach())
        y = torch.empty(b * ho * wo, co, dtype=torch.float32, device=input.device)
        d = H.make_conv_desc(b, [(h, w)], [(ho, wo)], [0], [0], c, co, co_pad, k, stride, padding, c, co,
                             flags=_lib.SM_CONV_OUT_F32, dil=dilation)
        H.conv2d(d, x, wq, None if bias is None else bias.detach().float().contiguous(), None, y)
        ctx.save_for_backward(x, weight)
        ctx.geom = (b, c, h, w, co, k, ho, wo, stride, padding, dilation, input.dtype, bias is not None)
        return y.view(b, ho, wo, co).permute(0, 3, 1, 2).to(input.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        x, weight = ctx.saved_tensors
        b, c, h, w, co, k, ho, wo, stride, pad, dil, dt, has_bias = ctx.geom
        dev = grad_output.device
        go = _rows_bf16(grad_output, co)
        d = H.make_conv_desc(b, [(h, w)], [(ho, wo)], [0], [0], c, co, co, k, stride, pad, c, co, dil=dil)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        K = k * k * c
        wdet = weight.detach()
        w_t = w_dg = None
        if need_x and stride == 1:       # flipped + transposed weight: dX is a forward conv over grad_output
            w_dg, _ = H.prep_conv_weight(wdet.flip(2, 3).permute(1, 0, 2, 3).contiguous(), co)
        elif need_x:                     # strided: grad columns + col2im
            w_t, _ = H.prep_conv_weight(wdet.permute(2, 3, 1, 0).reshape(K, co, 1, 1).contiguous(), co)
        gx = torch.empty(b * h * w, c, dtype=torch.float32, device=dev) if need_x else None
        gw_t = torch.empty(K, co, dtype=torch.float32, device=dev) if need_w else None
        gb = torch.empty(co, dtype=torch.float32, device=dev) if need_b else None
        H.conv2d_bwd(d, x, w_t, w_dg, go, gx, gw_t, gb)
        return (None if gx is None else gx.view(b, h, w, c).permute(0, 3, 1, 2).to(dt),
                None if gw_t is None else gw_t.view(k, k, c, co).permute(3, 2, 0, 1).contiguous().to(weight.dtype),
                gb, None, None, None)


conv2d = Conv2dFunction.apply


class GroupNormFunction(Function):
    """F.group_norm (+ fused ReLU) forward/backward on the HIP NCHW kernels (f32)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps=1e-5, relu=False):
        if not x.is_cuda:
            raise NotImplementedError
        xc = x.detach().float().contiguous()
        g, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y, stats = H.groupnorm_nchw_fwd(xc, g, b, groups, eps, relu)
        ctx.save_for_backward(xc, y, g, stats)
        ctx.cfg = (groups, relu, x.dtype)
        return y.to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y, g, stats = ctx.saved_tensors
        groups, relu, dt = ctx.cfg
        dx, dg, db = H.groupnorm_nchw_bwd(x, y, dy.detach().float().contiguous(), g, stats, groups, relu,
                                          ctx.needs_input_grad[0], ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return (None if dx is None else dx.to(dt), dg, db, None, None, None)


group_norm = GroupNormFunction.apply


class UpsampleBilinearFunction(Function):
    """F.interpolate(x, scale_factor=f, mode='bilinear', align_corners=False) for integer f, fwd/bwd in HIP."""

    @staticmethod
    def forward(ctx, x, factor):
        if not x.is_cuda:
            raise NotImplementedError
        ctx.factor, ctx.dt = int(factor), x.dtype
        return H.upsample_nchw(x.detach().float().contiguous(), int(factor)).to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return H.upsample_nchw(dy.detach().float().contiguous(), ctx.factor, backward=True).to(ctx.dt), None


upsample_bilinear = UpsampleBilinearFunction.apply


# ------------------------------------------------------------------------------- crop split
class CropSplitFunction(Function):

    @staticmethod
    def forward(ctx, data, rois, c):
        _lib.require_cuda(data, rois)
        if not data.is_contiguous():
            raise RuntimeError("data must be contiguous")   # crop_split_cuda.cpp:17
        if data.dtype != torch.float32:
            raise NotImplementedError("sipmask_amd crop_split: float32 only")
        height, width, n = data.shape[1], data.shape[2], data.shape[3]
        ctx.c, ctx.height, ctx.width, ctx.n = c, height, width, n
        ctx.save_for_backward(rois)
        output = torch.empty(height, width, n, dtype=data.dtype, device=data.device)
        lib = _lib.load()
        _lib.check(lib.sm_crop_split_fwd(_lib.ptr(data), _lib.ptr(rois.float().contiguous()), _lib.ptr(output),
                                         height, width, c, n, _lib.stream_ptr()), "sm_crop_split_fwd")
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        c, h, w, n = ctx.c, ctx.height, ctx.width, ctx.n
        grad_input = torch.empty((c * c, h, w, n), dtype=grad_output.dtype, device=grad_output.device)
        lib = _lib.load()
        _lib.check(lib.sm_crop_split_bwd(_lib.ptr(grad_output.contiguous()), _lib.ptr(rois.float().contiguous()),
                                         _lib.ptr(grad_input), h, w, c, n, _lib.stream_ptr()), "sm_crop_split_bwd")
        return grad_input, None, None


crop_split = CropSplitFunction.apply


class CropSplit(nn.Module):

    def __init__(self, c=2):
        super(CropSplit, self).__init__()
        self.c = c

    def forward(self, data, rois):
        return crop_split(data, rois, self.c)


class CropSplitGtFunction(Function):
    # no backward, as in the reference (crop_split_gt.py:11-27)

    @staticmethod
    def forward(ctx, data, rois, c):
        _lib.require_cuda(data, rois)
        if not data.is_contiguous():
            raise RuntimeError("data must be contiguous")
        height, width, n = data.shape
        output = torch.empty(height, width, n, dtype=data.dtype, device=data.device)
        lib = _lib.load()
        _lib.check(lib.sm_crop_split_gt_fwd(_lib.ptr(data), _lib.ptr(rois.float().contiguous()), _lib.ptr(output),
                                            height, width, n, _lib.stream_ptr()), "sm_crop_split_gt_fwd")
        return output


crop_split_gt = CropSplitGtFunction.apply


class CropSplitGt(nn.Module):

    def __init__(self, c=2):
        super(CropSplitGt, self).__init__()
        self.c = c

    def forward(self, data, rois):
        return crop_split_gt(data, rois, self.c)


# ------------------------------------------------------------------------------- focal loss
class SigmoidFocalLossFunction(Function):

    @staticmethod
    def forward(ctx, input, target, gamma=2.0, alpha=0.25):
        _lib.require_cuda(input, target)
        ctx.save_for_backward(input, target)
        num_classes = input.shape[1]
        ctx.num_classes, ctx.gamma, ctx.alpha = num_classes, gamma, alpha
        x = input.float().contiguous()
        loss = torch.empty_like(x)
        lib = _lib.load()
        _lib.check(lib.sm_sigmoid_focal_loss_fwd(_lib.ptr(x), _lib.ptr(target.long().contiguous()), _lib.ptr(loss),
                                                 x.shape[0], num_classes, gamma, alpha, _lib.stream_ptr()),
                   "sm_sigmoid_focal_loss_fwd")
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loss):
        input, target = ctx.saved_tensors
        x = input.float().contiguous()
        d_input = torch.empty_like(x)
        lib = _lib.load()
        _lib.check(lib.sm_sigmoid_focal_loss_bwd(_lib.ptr(x), _lib.ptr(target.long().contiguous()),
                                                 _lib.ptr(d_loss.float().contiguous()), _lib.ptr(d_input),
                                                 x.shape[0], ctx.num_classes, ctx.gamma, ctx.alpha,
                                                 _lib.stream_ptr()), "sm_sigmoid_focal_loss_bwd")
        return d_input, None, None, None


sigmoid_focal_loss = SigmoidFocalLossFunction.apply


# ------------------------------------------------------------------------------- fused mask loss
class MaskLossFunction(Function):
    """Per-detection sum of the mask BCE (sipmask_head.py:443-461) without the [4,Hm,Wm,N] volumes:
    S[n] = sum_pixels BCE(CropSplit(sigmoid(basis.cof_q))[..,n], CropSplitGt(gt[idx_gt[n]])[..,n]).
    feat_mask [32,Hm,Wm] f32, cof_pred [N,128], rois [N,4] (basis-grid boxes), gt_masks u8/bool/float 0-1
    [G,Hm,Wm], idx_gt long [N].  Differentiable in feat_mask and cof_pred."""

    @staticmethod
    def forward(ctx, feat_mask, cof_pred, rois, gt_masks, idx_gt):
        if not feat_mask.is_cuda:
            raise NotImplementedError
        basis = feat_mask.detach().float().contiguous()
        cof = cof_pred.detach().float().contiguous()
        rois = rois.detach().float().contiguous()
        gt = gt_masks.detach().to(torch.uint8).contiguous()
        idx = idx_gt.detach().long().contiguous()
        n = cof.shape[0]
        if basis.dim() != 3 or basis.shape[0] != 32 or cof.shape[1] != 128 or rois.shape != (n, 4) or \
                gt.shape[1:] != basis.shape[1:] or idx.shape != (n,):
            raise ValueError("mask_loss: inconsistent shapes")
        out = torch.empty(n, dtype=torch.float32, device=basis.device)      # zeroed by sm_mask_loss_fwd
        H.mask_loss_fwd(basis, cof, rois, gt, idx, out)
        ctx.save_for_backward(basis, cof, rois, gt, idx)
        ctx.dtypes = (feat_mask.dtype, cof_pred.dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_sum):
        basis, cof, rois, gt, idx = ctx.saved_tensors
        gb = torch.empty_like(basis) if ctx.needs_input_grad[0] else None
        gc = torch.empty_like(cof) if ctx.needs_input_grad[1] else None      # zeroed by sm_mask_loss_bwd
        H.mask_loss_bwd(basis, cof, rois, gt, idx, grad_sum.detach().float().contiguous(), gc, gb)
        return (None if gb is None else gb.to(ctx.dtypes[0]), None if gc is None else gc.to(ctx.dtypes[1]),
                None, None, None)


mask_loss = MaskLossFunction.apply


# ------------------------------------------------------------------------------- NMS
def nms(dets, iou_thr, device_id=None):
    """Same contract as nms_wrapper.nms for GPU tensors: returns (dets[inds], inds), inds ascending.
    Suppression rule IoU(+1) > iou_thr (the reference GPU kernel, nms_kernel.cu:61)."""
    if not isinstance(dets, torch.Tensor):
        raise TypeError('dets must be a Tensor, but got {}'.format(type(dets)))
    if dets.shape[0] == 0:
        return dets, dets.new_zeros(0, dtype=torch.long)
    _lib.require_cuda(dets)
    d = dets.float().contiguous()
    n = d.shape[0]
    keep = torch.empty(n, dtype=torch.int64, device=d.device)
    nkeep = torch.zeros(1, dtype=torch.int32, device=d.device)
    lib = _lib.load()
    _lib.check(lib.sm_nms(_lib.ptr(d), n, float(iou_thr), _lib.ptr(keep), _lib.ptr(nkeep), None, _lib.stream_ptr()),
               "sm_nms")
    inds = keep[:int(nkeep.item())]
    return dets[inds, :], inds


def multiclass_nms_idx(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """bbox_nms.py:79-146 on device for one image.  multi_scores [K, C+1] with a background column 0.
    Returns (bboxes [N,5], labels [N] long, idxs [N] long)."""
    _lib.require_cuda(multi_bboxes, multi_scores)
    if multi_bboxes.shape[1] != 4:
        raise NotImplementedError("class-specific boxes are not on the SipMask path")
    cfg = dict(nms_cfg)
    nms_type = cfg.pop('type', 'nms')
    if nms_type != 'nms':
        raise NotImplementedError("only greedy 'nms' (soft_nms is out of scope)")
    iou_thr = cfg.get('iou_thr', 0.5)
    k = multi_scores.shape[0]
    c = multi_scores.shape[1] - 1
    dev = multi_bboxes.device
    if k == 0:
        return (multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0,), dtype=torch.long),
                multi_bboxes.new_zeros((0,), dtype=torch.long))
    scores = multi_scores[:, 1:].float().t().contiguous().view(1, c, k)     # class-major for the kernel
    boxes = multi_bboxes.float().contiguous().view(1, k, 4)
    ctr = (torch.ones(1, k, device=dev) if score_factors is None else score_factors.float().contiguous().view(1, k))
    ncand = torch.full((1,), k, dtype=torch.int32, device=dev)
    # max_num <= 0 means "no cap" in the reference (bbox_nms.py:139-143): every (box, class) pair above score_thr can
    # survive, so the output is sized for all of them
    # the kernel's output list holds at most 2048 detections (TK_CAP, csrc/detect.hip)
    cap = max_num if max_num > 0 else min(k * c, 2048)
    out = H.multiclass_nms_alloc(1, k, c, cap, dev)
    H.multiclass_nms(boxes, scores, ctr, ncand, score_thr, iou_thr, cap, out)
    n = int(out["ndet"][0].item())
    if max_num <= 0 and n == cap and cap < k * c:
        # never truncate silently: an uncapped call that fills the kernel's list may have lost detections
        raise RuntimeError("multiclass_nms_idx(max_num=-1): %d detections survive NMS, the device list holds %d; pass max_num"
                           % (n, cap))
    return out["det"][0, :n], out["labels"][0, :n], out["keep"][0, :n]


def fast_nms(boxes, scores, masks, iou_threshold=0.5, top_k=200, score_thr=0.1, max_num=100):
    """SipMaskHead.fast_nms (sipmask_head.py:868-910) on device for one image.  boxes [K,4], scores [C,K]
    (already multiplied by centerness, :603), masks [K,D] (the coefficient rows).  Returns
    (boxes [N,5], classes [N] long, masks [N,D]); ties in a class are ranked by candidate index."""
    _lib.require_cuda(boxes, scores, masks)
    c, k = scores.shape
    dev = boxes.device
    if k == 0:
        return boxes.new_zeros((0, 5)), boxes.new_zeros((0,), dtype=torch.long), masks[:0]
    out = H.multiclass_nms_alloc(1, k, c, max_num, dev)
    H.fast_nms(boxes.float().contiguous().view(1, k, 4), scores.float().contiguous().view(1, c, k),
               torch.ones(1, k, device=dev), torch.full((1,), k, dtype=torch.int32, device=dev), score_thr,
               iou_threshold, top_k, max_num, out)
    n = int(out["ndet"][0].item())
    keep = out["keep"][0, :n]
    return out["det"][0, :n], out["labels"][0, :n], masks[keep]


def encode_masks(masks, ndet, canvas_hw, rect=None, max_runs=8192):
    """Device replacement of the reference's per-detection `mask.cpu()` + paste + `mask_util.encode` loop
    (sipmask_head.py:645-657).  masks u8 [B, max_num, ho, wo] (0/1), ndet int32 [B].  Returns, per image, the
    list of RLE dicts {'size': [H, W], 'counts': bytes} of its first ndet[b] detections.  The run buffer is
    grown and the launch repeated if a mask has more than max_runs runs."""
    _lib.require_cuda(masks, ndet)
    if masks.dtype != torch.uint8:
        raise TypeError("masks must be uint8 0/1")
    b, n = masks.shape[0], masks.shape[1]
    nd = ndet.cpu().numpy()
    while True:
        out = H.rle_alloc(b, n, int(canvas_hw[1]), masks.device, max_runs=max_runs)
        H.rle_encode(masks.contiguous(), ndet, canvas_hw, out, rect)
        nr = out["nruns"].min().item()
        if nr < 0:
            max_runs = int(-nr) + 1
            continue
        need = int(out["offsets"][-1].item())
        if need > out["packed"].numel():
            out = H.rle_alloc(b, n, int(canvas_hw[1]), masks.device, max_runs=max_runs, packed_cap=need)
            H.rle_encode(masks.contiguous(), ndet, canvas_hw, out, rect)
        return H.rle_fetch(out, b, n, nd, canvas_hw)


class Scale(nn.Module):
    """M/mmdet/ops/scale.py:5-15"""

    def __init__(self, scale=1.0):
        super(Scale, self).__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale
