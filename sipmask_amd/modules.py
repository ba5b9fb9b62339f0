"""Parameter containers with the reference's names, constructor kwargs and init rules.

These nn.Modules own the float32 parameters under exactly the reference ``state_dict`` keys
 so published checkpoints load; they contain NO arithmetic.  All compute runs
in the HIP engine (sipmask_amd/engine.py), which re-lays the weights out at ``prepare()`` time.

  ConvModule   M/mmdet/ops/conv_module.py:34-132  (norm attr name 'gn'/'bn': M/mmdet/ops/norm.py:7,40)
  ResNet       M/mmdet/models/backbones/resnet.py:311-521 (Bottleneck :84-239)
  FPN          M/mmdet/models/necks/fpn.py:10-178
"""
import math

import torch
import torch.nn as nn

from .registry import BACKBONES, NECKS


def kaiming_init(m, nonlinearity="relu"):
    nn.init.kaiming_normal_(m.weight, a=0, mode="fan_out", nonlinearity=nonlinearity)
    if getattr(m, "bias", None) is not None:
        nn.init.constant_(m.bias, 0)


def normal_init(m, mean=0, std=1, bias=0):
    nn.init.normal_(m.weight, mean, std)
    if getattr(m, "bias", None) is not None:
        nn.init.constant_(m.bias, bias)


def xavier_init(m, gain=1, bias=0):
    nn.init.xavier_uniform_(m.weight, gain=gain)
    if getattr(m, "bias", None) is not None:
        nn.init.constant_(m.bias, bias)


def bias_init_with_prob(prior_prob):
    """M/mmdet/models/utils/weight_init.py:4-7"""
    return float(-math.log((1 - prior_prob) / prior_prob))


def _norm_layer(cfg, channels):
    """(attr_name, module) like build_norm_layer: 'bn' / 'gn' prefixes, requires_grad handling."""
    cfg = dict(cfg)
    kind = cfg.pop("type")
    requires_grad = cfg.pop("requires_grad", True)
    if kind == "BN":
        name, layer = "bn", nn.BatchNorm2d(channels, eps=cfg.get("eps", 1e-5))
    elif kind == "GN":
        name, layer = "gn", nn.GroupNorm(cfg["num_groups"], channels, eps=cfg.get("eps", 1e-5))
    else:
        raise KeyError("Unrecognized norm type {}".format(kind))
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return name, layer


class ConvModule(nn.Module):
    """conv -> norm -> act parameter holder; bias only when there is no norm (conv_module.py:63-65)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias="auto",
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True,
                 order=("conv", "norm", "act")):
        super().__init__()
        assert conv_cfg is None, "only plain Conv2d is on the SipMask path"
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == "auto":
            bias = not self.with_norm
        self.with_bias = bias
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = self.conv.kernel_size, self.conv.stride, self.conv.padding
        self.norm_name = None
        if self.with_norm:
            self.norm_name, norm = _norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        kaiming_init(self.conv)
        if self.with_norm:
            nn.init.constant_(self.norm.weight, 1)
            nn.init.constant_(self.norm.bias, 0)

    @property
    def norm(self):
        return getattr(self, self.norm_name)


class DeformConvPack(nn.Module):
    """Parameter container of M/mmdet/ops/dcn/deform_conv.py:258-296 (conv type 'DCN'): ``weight`` of the
    deformable conv and the ordinary ``conv_offset`` conv (zero-initialised, :289-291) that predicts its offsets."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, deformable_groups=1):
        super().__init__()
        self.in_channels, self.out_channels, self.deformable_groups = in_channels, out_channels, deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.conv_offset = nn.Conv2d(in_channels, deformable_groups * 2 * kernel_size * kernel_size, kernel_size,
                                     stride=stride, padding=padding, bias=True)
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, style="pytorch", norm_cfg=dict(type="BN"),
                 dcn=None):
        super().__init__()
        assert style in ("pytorch", "caffe")
        self.conv1_stride, self.conv2_stride = (1, stride) if style == "pytorch" else (stride, 1)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=self.conv1_stride, bias=False)
        self.add_module("bn1", _norm_layer(norm_cfg, planes)[1])
        self.with_dcn = dcn is not None
        if self.with_dcn:                                     # resnet.py:145-168
            if dcn.get('type', 'DCN') != 'DCN' or dcn.get('fallback_on_stride', False):
                raise NotImplementedError("only conv type 'DCN' (deformable conv v1) is on the SipMask++ path")
            self.conv2 = DeformConvPack(planes, planes, 3, self.conv2_stride, 1, dcn.get('deformable_groups', 1))
        else:
            self.conv2 = nn.Conv2d(planes, planes, 3, stride=self.conv2_stride, padding=1, bias=False)
        self.add_module("bn2", _norm_layer(norm_cfg, planes)[1])
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.add_module("bn3", _norm_layer(norm_cfg, planes * 4)[1])
        self.downsample = downsample

    def forward_rows(self, x, lv):
        """resnet.py:205-239 on row tensors: three conv launches (+ the shortcut's), BN / ReLU / residual add fused"""
        from . import ops_rows as R
        out, l1 = _conv_bn_rows(x, lv, self.conv1, self.bn1, self.conv1_stride, 0, True)
        if self.with_dcn:
            n_off = self.conv2.conv_offset.weight.shape[0]
            wo, bo = _pad_cout8(self.conv2.conv_offset.weight, self.conv2.conv_offset.bias)
            off, _ = R.conv_rows(out, l1, wo, bo, 1, 1, out_f32=True)
            off = off[:, :n_off].contiguous()
            if self.bn2.weight.requires_grad or self.bn2.bias.requires_grad:
                w, b = _fold(self.conv2.weight, self.bn2)
                out = R.deform_conv_rows(out, l1, off, w, b, 1, 1, self.conv2.deformable_groups, relu=True)
            else:
                s2, b2 = R.bn_fold_constants(self.bn2)
                out = R.deform_conv_rows(out, l1, off, self.conv2.weight, b2, 1, 1, self.conv2.deformable_groups, relu=True,
                                         scale=s2)
            l2 = l1
        else:
            out, l2 = _conv_bn_rows(out, l1, self.conv2, self.bn2, self.conv2_stride, 1, True)
        idt = x if self.downsample is None else _conv_bn_rows(x, lv, self.downsample[0], self.downsample[1],
                                                              self.downsample[0].stride[0], 0, False)[0]
        return _conv_bn_rows(out, l2, self.conv3, self.bn3, 1, 0, True, residual=idt)

    def forward_train(self, x):
        """resnet.py:205-239 on the HIP autograd ops; the (eval-mode, frozen) BatchNorms are folded into the conv
        weights inside the graph, so their affine parameters would still receive gradients if they required them."""
        from . import ops as P
        out = torch.relu(_conv_bn(x, self.conv1, self.bn1, self.conv1_stride, 0))
        if self.with_dcn:
            wo, bo = _pad_cout8(self.conv2.conv_offset.weight, self.conv2.conv_offset.bias)
            off = P.conv2d(out, wo, bo, 1, 1)[:, :self.conv2.conv_offset.weight.shape[0]]
            w, b = _fold(self.conv2.weight, self.bn2)
            out = P.deform_conv(out, off.contiguous(), w, 1, 1, 1, 1, self.conv2.deformable_groups) + b.view(1, -1, 1, 1)
            out = torch.relu(out)
        else:
            out = torch.relu(_conv_bn(out, self.conv2, self.bn2, self.conv2_stride, 1))
        out = _conv_bn(out, self.conv3, self.bn3, 1, 0)
        idt = x if self.downsample is None else _conv_bn(x, self.downsample[0], self.downsample[1],
                                                         self.downsample[0].stride[0], 0)
        return torch.relu(out + idt)


def _train_rows_enabled(x):
    """the row-tensor training graph (ops_rows.py) is the default on the device; SIPMASK_TRAIN_ROWS=0 keeps the first
    version (per-op NCHW float interface, ops.py) for A/B runs, and CPU tensors take it too (the CPU emulation in
    tests/test_gpu_api.py swaps ops.conv2d under it)"""
    import os
    return x.is_cuda and os.environ.get("SIPMASK_TRAIN_ROWS", "1") != "0"


def _conv_bn_rows(x, lv, conv, bn, stride, pad, relu, residual=None):
    """conv + eval-mode BatchNorm (+ residual, + ReLU) as ONE conv launch on row tensors: with frozen affine parameters
    (the sipmask configs: norm_cfg requires_grad=False) the fold is a per-cout scale applied when the bf16 weight operand
    is written (sm_weight_prep) and undone on the weight gradient (sm_wgrad_finish); otherwise the differentiable
    tensor fold feeds the same op."""
    from . import ops_rows as R
    if bn.weight.requires_grad or bn.bias.requires_grad:
        w, b = _fold(conv.weight, bn)
        return R.conv_rows(x, lv, w, b, stride, pad, relu, None, residual)
    s, b = R.bn_fold_constants(bn)
    return R.conv_rows(x, lv, conv.weight, b, stride, pad, relu, s, residual)


def _fold(weight, bn):
    """eval-mode BatchNorm folded into the preceding conv: differentiable in weight (and in bn.weight/bias)."""
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return weight * s.view(-1, 1, 1, 1), bn.bias - bn.running_mean * s


def _conv_bn(x, conv, bn, stride, pad):
    from . import ops as P
    w, b = _fold(conv.weight, bn)
    return P.conv2d(x, w, b, stride, pad)


def _pad_cout8(w, b):
    """zero rows up to a multiple of 8 output channels (the GEMM kernels want cout % 8 == 0)"""
    extra = (-w.shape[0]) % 8
    if extra == 0:
        return w, b
    return torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, extra)), torch.nn.functional.pad(b, (0, extra))


@BACKBONES.register_module
class ResNet(nn.Module):
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, in_channels=3, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), style="pytorch", frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), with_cp=False, zero_init_residual=True):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError("invalid depth {} for resnet".format(depth))
        if conv_cfg is not None:
            raise NotImplementedError("conv_cfg other than the default Conv is not on the SipMask path")
        self.dcn, self.stage_with_dcn = dcn, tuple(stage_with_dcn)
        if style != "caffe" or tuple(strides) != (1, 2, 2, 2) or tuple(dilations) != (1, 1, 1, 1):
            raise NotImplementedError("the engine implements the caffe-style stride layout of the sipmask configs")
        self.depth, self.num_stages, self.out_indices = depth, num_stages, out_indices
        self.style, self.frozen_stages, self.norm_eval = style, frozen_stages, norm_eval
        self.zero_init_residual = zero_init_residual
        self.conv1 = nn.Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.add_module("bn1", _norm_layer(norm_cfg, 64)[1])
        inplanes = 64
        self.res_layers = []
        for i, nblocks in enumerate(self.arch_settings[depth][:num_stages]):
            planes = 64 * 2 ** i
            blocks = []
            for j in range(nblocks):
                stride = strides[i] if j == 0 else 1
                ds = None
                if j == 0 and (stride != 1 or inplanes != planes * 4):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                       _norm_layer(norm_cfg, planes * 4)[1])
                # SipMask++ edit of make_res_layer (resnet.py:270-291): DCN in block 0 and every 3rd block after it
                bdcn = dcn if (dcn is not None and self.stage_with_dcn[i] and j % 3 == 0) else None
                blocks.append(Bottleneck(inplanes, planes, stride, ds, style, norm_cfg, bdcn))
                inplanes = planes * 4
            name = "layer{}".format(i + 1)
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        self._freeze_stages()

    def forward_rows(self, img):
        """resnet.py:501-512 on row tensors: list of (rows bf16 [B*h*w, C], Levels) per out_index.  Stem and frozen
        stages run without a graph; BN is always in eval mode (norm_eval, cfg requires_grad=False)."""
        from . import ops_rows as R
        from . import hip_ops as H
        if not self.norm_eval or self.frozen_stages < 0:
            raise NotImplementedError("ResNet.forward_rows implements the sipmask configs: norm_eval=True and "
                                      "frozen_stages >= 0 (got norm_eval=%r, frozen_stages=%r)" % (self.norm_eval, self.frozen_stages))
        b, _, hh, ww = img.shape
        outs = []
        with torch.no_grad():
            x = R.nchw_to_rows(img, 8)
            x, lv = _conv_bn_rows(x, H.Levels(b, [(hh, ww)]), self.conv1, self.bn1, 2, 3, True)
            h, w = lv.sizes[0]
            ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
            y = torch.empty(b * ho * wo, 64, dtype=torch.bfloat16, device=img.device)
            H.maxpool3x3s2(x, y, b, h, w, 64)
            x, lv = y, H.Levels(b, [(ho, wo)])
        for i, name in enumerate(self.res_layers):
            frozen = (i + 1) <= self.frozen_stages
            with torch.set_grad_enabled(torch.is_grad_enabled() and not frozen):
                for blk in getattr(self, name):
                    x, lv = blk.forward_rows(x, lv)
            if frozen:
                x = x.detach()
            if i in self.out_indices:
                outs.append((x, lv))
        return outs

    def forward_train(self, x):
        """resnet.py:501-512 as a differentiable graph (BN always in eval mode: norm_eval, cfg requires_grad=False).
        Frozen stages run without building a graph.  x: [B,3,H,W] float on the device."""
        from . import ops as P
        if not self.norm_eval or self.frozen_stages < 0:
            # the training graph folds eval-mode BatchNorm into the convs and runs the stem without a graph
            raise NotImplementedError("ResNet.forward_train implements the sipmask configs: norm_eval=True and "
                                      "frozen_stages >= 0 (got norm_eval=%r, frozen_stages=%r)" % (self.norm_eval, self.frozen_stages))
        if _train_rows_enabled(x):
            from .ops_rows import rows_to_nchw
            return tuple(rows_to_nchw(r, lv.batch, *lv.sizes[0]) for r, lv in self.forward_rows(x))
        outs = []
        with torch.no_grad():
            x8 = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, 5))                      # 3 -> 8 input channels
            w, b = _fold(torch.nn.functional.pad(self.conv1.weight, (0, 0, 0, 0, 0, 5)), self.bn1)
            x = torch.relu(P.conv2d(x8, w, b, 2, 3))
            x = torch.nn.functional.max_pool2d(x, 3, 2, 1)
        for i, name in enumerate(self.res_layers):
            frozen = (i + 1) <= self.frozen_stages
            with torch.set_grad_enabled(torch.is_grad_enabled() and not frozen):
                for blk in getattr(self, name):
                    x = blk.forward_train(x)
            if frozen:
                x = x.detach()
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, "layer{}".format(i))
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            sd = torch.load(pretrained, map_location="cpu")
            self.load_state_dict(sd.get("state_dict", sd), strict=False)
        elif pretrained is None:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    kaiming_init(m)
                elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                    nn.init.constant_(m.weight, 1)
                    nn.init.constant_(m.bias, 0)
            if self.zero_init_residual:
                for m in self.modules():
                    if isinstance(m, Bottleneck):
                        nn.init.constant_(m.bn3.weight, 0)
        else:
            raise TypeError("pretrained must be a str or None")

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()


@NECKS.register_module
class FPN(nn.Module):

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None):
        super().__init__()
        assert isinstance(in_channels, list)
        if norm_cfg is not None or conv_cfg is not None or end_level != -1:
            raise NotImplementedError("FPN variants outside the sipmask configs")
        if not (add_extra_convs and not extra_convs_on_inputs and relu_before_extra_convs):
            raise NotImplementedError("the engine implements extra convs on outputs with ReLU (sipmask cfg :13-21)")
        self.in_channels, self.out_channels, self.num_outs = in_channels, out_channels, num_outs
        self.num_ins, self.start_level = len(in_channels), start_level
        self.backbone_end_level = self.num_ins
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, act_cfg=act_cfg, inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, act_cfg=act_cfg, inplace=False))
        for i in range(num_outs - self.backbone_end_level + start_level):
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, stride=2, padding=1, act_cfg=act_cfg,
                                             inplace=False))

    def forward_rows(self, inputs):
        """fpn.py:137-178 on row tensors: inputs = [(rows, Levels)] per backbone level -> [(rows, Levels)] per output.
        The top-down nearest-neighbour add rides in the lateral conv's epilogue (SM_CONV_RES_NEAREST)."""
        from . import ops_rows as R
        n = len(self.lateral_convs)
        lats = [None] * n
        for i in range(n - 1, -1, -1):
            x, lv = inputs[i + self.start_level]
            lc = self.lateral_convs[i].conv
            if i == n - 1:
                lats[i] = R.conv_rows(x, lv, lc.weight, lc.bias, 1, 0)
            else:
                lats[i] = R.conv_rows(x, lv, lc.weight, lc.bias, 1, 0, residual=lats[i + 1][0], res_mode='nearest',
                                      res_lv=lats[i + 1][1])
        outs = [R.conv_rows(lats[i][0], lats[i][1], self.fpn_convs[i].conv.weight, self.fpn_convs[i].conv.bias, 1, 1)
                for i in range(n)]
        for i in range(n, len(self.fpn_convs)):
            src, lv = outs[-1]
            if i > n:
                src = torch.relu(src)
            outs.append(R.conv_rows(src, lv, self.fpn_convs[i].conv.weight, self.fpn_convs[i].conv.bias, 2, 1))
        return outs

    def forward_train(self, inputs):
        """fpn.py:137-178 (start_level, extra convs on the outputs, ReLU before P7) on the HIP conv autograd op."""
        from . import ops as P
        lats = [P.conv2d(inputs[i + self.start_level], lc.conv.weight, lc.conv.bias, 1, 0)
                for i, lc in enumerate(self.lateral_convs)]
        for i in range(len(lats) - 1, 0, -1):
            lats[i - 1] = lats[i - 1] + torch.nn.functional.interpolate(lats[i], size=lats[i - 1].shape[2:], mode='nearest')
        outs = [P.conv2d(lats[i], self.fpn_convs[i].conv.weight, self.fpn_convs[i].conv.bias, 1, 1)
                for i in range(len(lats))]
        for i in range(len(lats), len(self.fpn_convs)):
            src = outs[-1] if i == len(lats) else torch.relu(outs[-1])
            outs.append(P.conv2d(src, self.fpn_convs[i].conv.weight, self.fpn_convs[i].conv.bias, 2, 1))
        return tuple(outs)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m)
