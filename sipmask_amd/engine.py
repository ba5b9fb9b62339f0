"""Inference engine: the SipMask hot path as a static plan of HIP kernel launches.

``SipMaskEngine.prepare`` takes a reference-named ``state_dict``, folds the
frozen BatchNorms into the conv weights (resnet.py:370,514-521: BN is always eval), re-lays the
weights out as K-major bf16 GEMM operands, allocates every activation / workspace buffer once
(288 GB of HBM: nothing is ever re-allocated) and records the launch list.  ``run`` replays the
list on the current stream -- it is capture-safe, so callers may wrap it in a HIP graph.

Layout: activations are NHWC bf16 "pyramid tensors" (all FPN levels of all images in one row
matrix) so the 5 levels that share tower weights run as ONE implicit-GEMM launch.
"""
import math
import os

import torch

from . import _lib
from . import hip_ops as H
from ._lib import SM_CONV_RELU, SM_CONV_OUT_F32, SM_CONV_RES_ADD, SM_CONV_RES_NEAREST, SM_CONV_IN_RELU

ARCH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

# ---- environment switches of the engine (INTEGRATION.md section 4 lists every SIPMASK_* variable of the package) -----------
# launch-plan selector bits of include/sipmask_hip.h OR-ed into every conv descriptor (whole-plan experiments, tools/)
_DEBUG_CONV_FLAGS = int(os.environ.get("SIPMASK_CONV_DEBUG_FLAGS", "0"), 0)
# FeatureAlign's deformable conv: the LDS-window kernel (csrc/deform_patch.hip) is 1.3-1.6x the gather loader while the
# learned offsets stay within ~3 pixels and falls behind it when most waves sample farther out (random offsets of sigma 4:
# 0.35 vs 0.26 ms at B=4, profiles/r02_deform_conv_microbench.txt).  "auto" (default) inspects the offsets of the first
# eager run of the plan (the checkpoint's own offsets on a real input) and keeps the gather loader where more than 20 % of
# the offset components reach beyond the window radius (SipMaskEngine._tune_deform); "0" / "1" pin the window kernel / the
# gather loader.
_DEFORM_MODE = os.environ.get("SIPMASK_DEFORM_GATHER", "auto")
_DEFORM_FLAGS = _lib.SM_CONV_DBG_DEFORM_GATHER if _DEFORM_MODE == "1" else 0
# FeatureAlign's deformable conv in the x3 head plan: "window" (default) = the LDS-window kernel csrc/deform_patch_x3.hip where
# the shape is its own and the offsets stay near (the rule above); "f32x3" = f32 rows, operands split in the gather loader,
# f16 MFMAs (the round-3 kernel); "f32" = the exact-f32 MFMA kernel
_X3_FEAT_ALIGN = os.environ.get("SIPMASK_X3_FEAT_ALIGN", "window")
# diagnostic (tools/marginal_cost.sh): launches whose label matches are left out of CAPTURED graphs; bench.py marks its line
_DIAG_SKIP = __import__("re").compile(os.environ["SIPMASK_DIAG_SKIP"]) if os.environ.get("SIPMASK_DIAG_SKIP") else None

# ---- plan-structure constants --------------------------------------------------------------------------------------------
# Each of these was an environment A/B switch while it was being measured (rounds 2-4; HISTORY.md / DESIGN.md section 6 hold
# the numbers); the measured best is fixed here and the variables are gone (41 switches were 41 untested
# configurations of the shipped path).  tests/test_gpu_engine.py monkeypatches them to hold the alternative launch structures
# to bit-identical results, which is what they remain good for.
_GROUPED_TOWERS = True         # cls + reg tower convs of one depth as ONE grouped 256x256-tile launch (950 vs 785 TFLOP/s)
_SPLIT_K = True                # split-K for under-filled launches of lone plans
_SPLIT_K_SMALL_FPN = True      # ... and of lat2 / P6 / P7 inside pipelined slots
_PATCH_CONV = True             # the patch-resident 3x3 kernel (csrc/conv3x3_patch.hip)
_PATCH_COUT128 = True          # its 128-cout tile (layer3 / layer4 conv2)
_PATCH_SMALL_COUT = True       # its 32-cout tile (where the small-cout kernel below does not take the conv)
_PATCH_MIN_WORK = 100.0        # 256x256 tile equivalents a launch must have to take the patch kernel ...
_PATCH_MIN_FILL = 0.6          # ... and the share of the CUs its planned shape keeps busy
# bottleneck fusion in layer1 / layer2: 0 = separate launches, 1 = conv2+conv3, 2 = conv2+conv3+next conv1 everywhere
# (profiles/r02f_ab_bottleneck_fusion.json: 963 / 998 / 984 img/s)
_FUSE_BOTTLENECK = 1
_CHAIN_CONV1 = 1               # layer1 tails also compute the next block's conv1 (2: not behind the fused shortcut; 3: layer2 too)
_FUSE_SHORTCUT = 2             # the shortcut conv inside the fused tail: 1 = layer1.0, 2 = + layer2.0
_PAIR_1X1 = False              # layer3's conv3 + next conv1 as one launch (sm_conv1x1_pair): bit-identical, no faster
_X3_TOWER0_TWO_TERMS = True    # x3 plan: the first tower convs on [hi | hi] of the bf16 FPN outputs (two half products per element)
# x3 plan, round 6: the 3x3 tower convs and fcos_cls + sip_cof read PAIRED split operands (per 16 channels [hi 16 | lo 16],
# sm_conv_desc.x3_pairs: three products on fragments read once, a third less LDS-DMA traffic than [hi | lo | hi] x [hi | hi | lo])
_X3_PAIRS = True
_LATENCY_1X1 = ()              # stage widths whose 1x1 convs keep the latency-shaped plan inside pipelined slots: neutral
_RELU_COPY_P7 = True           # relu(P6) as its own tensor instead of the input-ReLU loader for P7
_SMALLCO_CONV = True           # 3x3 convs with <= 32 couts on csrc/conv3x3_smallco.hip
_STEM_FUSED = True             # conv1 + bn1 + relu + maxpool as one launch (csrc/stem_fused.hip)
_LAT0_LINEAR = True            # sip_mask_lat0 by linearity (three convs + sm_upsample_sum2)
_FUSED_MASKS = True            # coefficient x basis, x4 upsample, sigmoid, threshold, crop in one kernel
# FPN output convs of levels 0-2 as ONE launch with per-level weights (_LevelConv): "auto" = where the patch kernel takes the
# shape and the launch is at least _LEVEL_CONV_MIN_WORK tile equivalents, "1" = wherever it is supported (tests at small
# shapes), "0" = three launches
_FPN_GROUPED = "auto"
_LEVEL_CONV_MIN_WORK = 50.0
_PATCH_MIXED_IN_CHAINS = False     # mixed 256 / 128 / 192-position tiles inside SubBatchPlan chains (998 vs 991 img/s uniform)
_PATCH_UNIFORM_IN_SLOTS = False    # uniform 256-position tiles inside PipelinedPlan slots (slower)


def _lib_flag(name):
    return getattr(_lib, name)
BF16 = torch.bfloat16


def _conv_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def fold_bn(w, sd, p, eps=1e-5):
    """conv (no bias) followed by eval-mode BN -> (w', b')."""
    g, b = sd[p + ".weight"].float(), sd[p + ".bias"].float()
    m, v = sd[p + ".running_mean"].float(), sd[p + ".running_var"].float()
    s = g / torch.sqrt(v + eps)
    return w.float() * s.view(-1, 1, 1, 1), b - m * s


class _Conv:
    """One prepared conv launch.  mode: "bf16" (bf16 operands, the throughput kernels), "f32" (exact-f32 MFMA kernel,
    csrc/conv_f32.hip) or "x3" (split-precision: binary16 halves [hi | lo | hi] of f32 activations against weights
    [hi | hi | lo] on the bf16 plan's own kernels with the f16 MFMA, SM_CONV_F16 -- `x` is then the 3*ci-channel split
    tensor, `y` f32).  Default: the plan's precision ("f32" plan -> "f32", else "bf16")."""

    def __init__(self, eng, name, w, bias, batch, in_sizes, in_row0, x, in_cstride, stride, pad, y, out_row0,
                 out_cstride, out_coff=0, flags=0, cin_pad=None, residual=None, res_cstride=0, res_sizes=None,
                 res_row0=None, scale_nch=0, level_scale=None, deform_groups=0, offset=None, mode=None, split_k=None):
        dev = eng.device
        co, ci, k, _ = w.shape
        cin = cin_pad or ci
        self.name = name
        self.mode = mode or ("f32" if getattr(eng, "precision", "bf16") == "f32" else "bf16")
        # exact-f32 plan (parity mode): f32 operands on v_mfma_f32_32x32x2_f32, every conv output f32
        self.f32 = self.mode in ("f32", "f32x3")            # f32 tensors in HBM: csrc/conv_f32.hip
        self.x3 = self.mode in ("x3", "x2", "x3p")           # "x2": the two-term form for inputs whose low half is zero
        self.x3p = self.mode == "x3p"                        # paired operands [hi 16 | lo 16] per 16 channels (patch kernel only)
        self.xterms = 2 if self.mode == "x2" else 3
        self.x3w = self.mode == "x3w"                        # FeatureAlign in the x3 plan on the LDS-window kernel
        acc_scale = 0.0
        if self.x3w:
            # f32 rows in / out, [hi | lo] binary16 weights per K step (csrc/deform_patch_x3.hip, round 5)
            if (offset is None or residual is not None or cin_pad is not None or k != 3 or stride != 1 or pad != 1
                    or ci != 64 * deform_groups):
                raise NotImplementedError("x3w: FeatureAlign's deformable 3x3 conv (64 channels per deformable group)")
            self.x3_scale = H.x3_weight_scale([w])
            self.w, co_pad = H.prep_deform_weight_x3(w.to(dev), self.x3_scale, deform_groups)
            flags |= _lib.SM_CONV_F16 | SM_CONV_OUT_F32
            acc_scale = 1.0 / self.x3_scale
        elif self.f32:
            cin = cin_pad = ci if ci % 4 == 0 else (ci + 3) // 4 * 4
            if self.mode == "f32x3":
                # the same kernel with the contraction in split precision (operands split into binary16 halves in the
                # loader, three f16 MFMAs per product): FeatureAlign's deformable conv in the x3 head plan
                self.x3_scale = H.x3_weight_scale([w])
                self.w, co_pad = H.prep_conv_weight_f32(w.to(dev).float() * self.x3_scale, cin)
                flags |= _lib.SM_CONV_F16
                acc_scale = 1.0 / self.x3_scale
            else:
                self.w, co_pad = H.prep_conv_weight_f32(w.to(dev), cin)
        elif self.x3p:
            if offset is not None or residual is not None or cin_pad is not None or ci % 16 != 0 or stride != 1:
                raise NotImplementedError("x3p convs: plain stride-1 convolutions over 16-aligned channel counts")
            cin = 2 * ci
            assert in_cstride == cin, "x3p convs read the paired split tensor (2 * cin binary16 per row)"
            self.x3_scale = getattr(self, "_x3_scale", None) or H.x3_weight_scale([w])
            # three kernels take paired operands: the patch-resident 3x3 kernel (256-cout tiles), the small-cout 3x3 kernel and
            # the implicit-GEMM kernel's 32-wide-K loop (1x1 convs); the weights below are the last one's, replaced further down
            self.w, co_pad = H.prep_conv_weight_x3p(w.to(dev), self.x3_scale)
            flags |= _lib.SM_CONV_F16 | SM_CONV_OUT_F32
            acc_scale = 1.0 / self.x3_scale
            self._patch_force = k == 3 and pad == 1 and co > 32
        elif self.x3:
            if offset is not None or residual is not None or cin_pad is not None or ci % 8 != 0:
                raise NotImplementedError("x3 convs: plain convolutions over 8-aligned channel counts")
            cin = self.xterms * ci
            assert in_cstride == cin, "x3 convs read the [hi | lo | hi] split tensor (3 * cin channels; x2: [hi | hi])"
            self.x3_scale = getattr(self, "_x3_scale", None) or H.x3_weight_scale([w])
            self.w, co_pad = H.prep_conv_weight_x3(w.to(dev), self.x3_scale, self.xterms)
            flags |= _lib.SM_CONV_F16 | (0 if flags & _lib.SM_CONV_OUT_X3 else SM_CONV_OUT_F32)
            acc_scale = 1.0 / self.x3_scale
        else:
            self.w, co_pad = H.prep_conv_weight(w.to(dev), cin)
        self.bias = None if bias is None else bias.float().to(dev).contiguous()
        out_sizes = [(_conv_out(h, k, stride, pad), _conv_out(ww, k, stride, pad)) for h, ww in in_sizes]
        self.out_sizes = out_sizes
        flags |= _DEBUG_CONV_FLAGS          # plan selectors of include/sipmask_hip.h for whole-plan experiments
        if not (split_k is True and _SPLIT_K_SMALL_FPN and _SPLIT_K):   # ... and the engine's own (PipelinedPlan slots: big tiles), except
            flags |= getattr(eng, "extra_conv_flags", 0)   # for the launches that keep the latency-shaped plan (below)
        self.desc = H.make_conv_desc(batch, in_sizes, out_sizes, in_row0, out_row0, cin, co, co_pad, k, stride, pad,
                                     in_cstride, out_cstride, out_coff, flags, 1, res_cstride, res_sizes, res_row0,
                                     scale_nch, level_scale, deform_groups, acc_scale=acc_scale, x3_pairs=int(self.x3p))
        self.x, self.y, self.residual, self.offset = x, y, residual, offset
        self.gn_stats = None      # set -> GroupNorm statistics of y are accumulated in the conv epilogue
        # large 3x3 / stride-1 convs run on the patch-resident kernel (csrc/conv3x3_patch.hip): own weight layout, cout
        # padded to 256.  `groups` launches of this shape share one grid (tower pairs): the tile rule sees all of them.
        self.patch = False
        # 3x3 convs with a handful of output channels on the bf16 plan (sip_mask_lat 512 -> 32, fcos_reg + centerness 256 -> 8):
        # their own kernel (round 4, csrc/conv3x3_smallco.hip: one wave per 2 x 32-position tile, weights straight from L2)
        self.smallco = False
        if (not self.f32 and offset is None and residual is None and _SMALLCO_CONV and k == 3 and stride == 1
                and pad == 1 and co <= 32 and co % 8 == 0 and (ci % 32 == 0 or self.x3p)
                and cin == (2 * ci if self.x3p else (3 * ci if self.x3 else ci)) and self.mode != "x2"
                and getattr(self, "_patch_groups", 1) == 1 and not (self.x3 and (flags & _lib.SM_CONV_OUT_X3))):
            ds = H.make_conv_desc(batch, in_sizes, out_sizes, in_row0, out_row0, cin, co, 32, k, stride, pad, in_cstride,
                                  out_cstride, out_coff, flags, 1, res_cstride, res_sizes, res_row0, scale_nch, level_scale,
                                  deform_groups, acc_scale=acc_scale, x3_pairs=int(self.x3p))
            if H.conv3x3_smallco_supported(ds):
                self.smallco = True
                self.w = H.prep_conv_weight_smallco(w.to(dev), x3_scale=self.x3_scale if self.x3 else None, pairs=self.x3p)
                self.desc = ds
        if (not self.smallco and not self.f32 and offset is None and residual is None and _PATCH_CONV and k == 3 and stride == 1
                and pad == 1 and (ci % 64 == 0 or (self.x3p and ci % 32 == 0)) and (cin == ci or self.x3)):
            # cout tile of the patch kernel: 256, or 32 for the convs with a handful of output channels (round 4: sip_mask_lat
            # 512 -> 32 and fcos_reg + centerness 256 -> 8 spent 0.10 / 0.065 ms per B=4 launch on the implicit-GEMM kernel
            # re-reading their INPUT nine times; the grouped / per-level launches keep the 256 tile)
            small_co = co <= 32 and co % 8 == 0 and getattr(self, "_patch_groups", 1) == 1 and _PATCH_SMALL_COUT and not self.x3p
            pad_co = H.patch_cout_pad(co) if small_co else (co + 255) // 256 * 256
            dp = H.make_conv_desc(batch, in_sizes, out_sizes, in_row0, out_row0, cin, co, pad_co, k, stride,
                                  pad, in_cstride, out_cstride, out_coff, flags, 1, res_cstride, res_sizes, res_row0,
                                  scale_nch, level_scale, deform_groups, acc_scale=acc_scale, x3_pairs=int(self.x3p))
            if H.conv3x3_patch_supported(dp):
                if getattr(eng, "patch_uniform", False):
                    dp.flags |= _lib.SM_CONV_DBG_PATCH_UNIFORM
                dp.ngroups = getattr(self, "_patch_groups", 1)       # the launch shape depends on every group's tiles
                pl = H.conv3x3_patch_plan(dp)
                dp.ngroups = 1
                # one tile per CU; the library cuts the launch into 256-position tiles + smaller finishing tiles
                # (sm_conv3x3_patch_plan).  Take the kernel when the launch is a reasonable share of a round of 256 CUs,
                # the planned shape keeps the CUs busy (fill = work / (256 x makespan)) and the couts fill most of the
                # 256-wide tile (sip_mask_lat, 128 -> 32: 0.061 ms on the 32 x 256 implicit-GEMM tile, 0.101 ms here with
                # 7/8 of the MFMAs on padding).  Measured: profiles/r02d_patch_conv_microbench.txt, r02g_patch_tile_shapes_microbench.txt.
                force = getattr(self, "_patch_force", False)
                # (small_co: the alternative is the implicit-GEMM kernel's nine-fold re-read of the input, whatever the fill)
                if ((force or (pl["work"] >= getattr(self, "_patch_min_work", _PATCH_MIN_WORK) and
                               (small_co or pl["fill"] >= _PATCH_MIN_FILL or pl["makespan"] <= 1.0)))
                        and (small_co or force or co * 4 >= 3 * ((co + 255) // 256 * 256))):
                    self.patch = True
                    self.w, _ = (H.prep_conv_weight_patch_x3p(w.to(dev), self.x3_scale, pad_co) if self.x3p else
                                 H.prep_conv_weight_patch_x3(w.to(dev), self.x3_scale, pad_co, self.xterms) if self.x3
                                 else H.prep_conv_weight_patch(w.to(dev), pad_co))
                    self.desc = dp
        if self.x3p and k == 3 and not (self.patch or self.smallco):
            raise NotImplementedError("x3p conv %s: neither 3x3 kernel with paired operands takes this shape" % name)
        # 128-cout x 256-position patch tiles (round 4) for single-level 3x3 convs whose position count gives too few 256-cout
        # tiles to be worth a launch of them: ResNet layer3 / layer4 conv2 (4 200 / 1 050 positions per image: 66 / 17 position
        # tiles at B=4, x 2 / 4 cout tiles).  The implicit-GEMM kernel re-reads their input nine times through L2 -> LDS and is
        # bound by it (HISTORY "stream-K ... analysed"): 0.045 / 0.075 ms per launch.  The rule is per IMAGE (h * w >= 900), so
        # that plans of different batch cuts choose alike (plan-to-plan bit equality).
        if (not self.patch and not self.f32 and not self.x3 and offset is None and residual is None and _PATCH_CONV
                and _PATCH_COUT128 and k == 3 and stride == 1 and pad == 1 and ci % 64 == 0 and cin == ci and co % 128 == 0
                and len(in_sizes) == 1 and getattr(self, "_patch_groups", 1) == 1 and in_sizes[0][0] * in_sizes[0][1] >= 900):
            dp = H.make_conv_desc(batch, in_sizes, out_sizes, in_row0, out_row0, cin, co, co, k, stride, pad, in_cstride,
                                  out_cstride, out_coff, flags, 1, res_cstride, res_sizes, res_row0, scale_nch, level_scale,
                                  deform_groups, acc_scale=acc_scale, patch_cout_tile=128)
            if H.conv3x3_patch_supported(dp):
                self.patch = True
                self.w, _ = H.prep_conv_weight_patch(w.to(dev), co)
                self.desc = dp
        # split-K workspace (own buffer per conv: launches on different lanes may run concurrently); sized by the
        # library's plan, allocated once at build -- 288 GB of HBM
        self.ws = None
        # (split_k: None = the engine's rule; True = also inside a pipelined slot -- the FPN's three launch-latency-shaped
        # convs, lat2 / P6 / P7: 6-66 tiles with 32-36 serial K steps, where the reduce launch costs no CU time worth counting)
        want_split = getattr(eng, "split_k", _SPLIT_K) if split_k is None else (bool(split_k) and _SPLIT_K_SMALL_FPN and _SPLIT_K)
        if not self.f32 and offset is None and want_split and not self.patch and not self.smallco:
            pl = H.conv_plan(self.desc)
            if pl["split_k"] > 1:
                self.ws = torch.empty(pl["workspace_bytes"], dtype=torch.uint8, device=dev)
        kin = self.xterms if self.x3 else (3 if self.mode in ("f32x3", "x3w") else 1)   # MFMA work: half products per element product
        self.flops = 2.0 * sum(batch * h * ww for h, ww in out_sizes) * co * ci * k * k
        self.mfma_flops = self.flops * kin
        # algorithmic HBM bytes: x read once + y written once (+ residual read) + weights
        in_rows = sum(batch * h * ww for h, ww in in_sizes)
        out_rows = sum(batch * h * ww for h, ww in out_sizes)
        es = 4 if (self.f32 or self.x3w) else 2
        self.bytes = (in_rows * cin * es + out_rows * co * (4 if (flags & SM_CONV_OUT_F32 or self.f32) else 2) +
                      (out_rows * co * es if residual is not None else 0) + self.w.numel() * (2 if self.x3w else es))

    def __call__(self):
        if self.x3w:
            H.deform_conv2d_x3(self.desc, self.x, self.offset, self.w, self.bias, self.y, self.gn_stats)
        elif self.f32:
            H.conv2d_f32(self.desc, self.x, self.offset, self.w, self.bias, self.residual, self.y)
        elif self.smallco:
            H.conv3x3_smallco(self.desc, self.x, self.w, self.bias, self.y)
        elif self.patch:
            H.conv3x3_patch(self.desc, self.x, self.w, self.bias, self.y, self.gn_stats)
        elif self.gn_stats is not None:
            H.conv2d_gn_stats(self.desc, self.x, self.offset, self.w, self.bias, self.residual, self.y, self.gn_stats)
        elif self.offset is not None:
            H.deform_conv2d(self.desc, self.x, self.offset, self.w, self.bias, self.y)
        elif self.ws is not None:
            H.conv2d_ws(self.desc, self.x, self.w, self.bias, self.residual, self.y, self.ws)
        else:
            H.conv2d(self.desc, self.x, self.w, self.bias, self.residual, self.y)


class _BottleneckTail:
    """conv2 + conv3 (+ the next block's conv1) of a ResNet bottleneck as ONE launch (csrc/bottleneck.hip;
    resnet.py:167-200): the C-channel tensor between conv2 and conv3, and the read of the block output by the next
    conv1, never touch HBM.  Same arithmetic as the separate launches (bit-identical outputs)."""

    def __init__(self, name, batch, hw, planes, x, w2, b2, w3, b3, identity, y, next1=None, shortcut=None):
        dev = x.device
        self.name, self.batch, self.hw, self.planes = name, batch, hw, planes
        prep = lambda w: H.prep_conv_weight(w.to(dev), w.shape[1])[0][:w.shape[0]].contiguous()
        fb = lambda b: b.float().to(dev).contiguous()
        self.x, self.identity, self.y = x, identity, y
        self.w2, self.b2, self.w3, self.b3 = prep(w2), fb(b2), prep(w3), fb(b3)
        self.w1n = self.b1n = self.t1n = None
        rows = batch * hw[0] * hw[1]
        self.flops = 2.0 * rows * planes * planes * 9 + 2.0 * rows * planes * 4 * planes
        # algorithmic bytes of the fused launch: x in, identity in, y out (+ next t1 out) + weights
        self.bytes = rows * planes * 2 + 2 * rows * 4 * planes * 2 + (self.w2.numel() + self.w3.numel()) * 2
        # shortcut = (w_downsample, b_downsample, block input rows): the block's 1x1 shortcut conv rides conv3's K loop
        # (layer1's first block: resnet.py:453-469; csrc/bottleneck.hip CDS) -- `identity` is not used
        self.x_block = None
        self.ds_stride, self.ds_hw = 1, hw
        if shortcut is not None:
            wd, bd, self.x_block = shortcut[:3]
            if len(shortcut) > 3:
                self.ds_stride, self.ds_hw = shortcut[3], shortcut[4]
            self.w3 = torch.cat([self.w3, prep(wd)], 1).contiguous()
            self.b3 = (self.b3 + fb(bd)).contiguous()
            cds = self.x_block.shape[1]
            self.flops += 2.0 * rows * cds * 4 * planes
            self.bytes = rows * planes * 2 + rows * cds * 2 + rows * 4 * planes * 2 + (self.w2.numel() + self.w3.numel()) * 2
        if next1 is not None:
            w1, b1, t1n = next1
            self.w1n, self.b1n, self.t1n = prep(w1), fb(b1), t1n
            self.flops += 2.0 * rows * 4 * planes * planes
            self.bytes += rows * planes * 2 + self.w1n.numel() * 2

    def __call__(self):
        if self.x_block is not None:
            H.bottleneck_tail_ds(self.batch, self.hw[0], self.hw[1], self.planes, self.x, self.w2, self.b2, self.w3, self.b3,
                                 self.x_block, self.y, self.w1n, self.b1n, self.t1n, self.ds_stride, self.ds_hw)
            return
        H.bottleneck_tail(self.batch, self.hw[0], self.hw[1], self.planes, self.x, self.w2, self.b2, self.w3, self.b3,
                          self.identity, self.y, self.w1n, self.b1n, self.t1n)


class _Conv1x1Pair:
    """conv3 (+ identity, ReLU) of one bottleneck and conv1 (ReLU) of the next as ONE launch (csrc/bottleneck.hip, CONV2 = false;
    resnet.py:188-200 then :175-178): ResNet layer3, where each of the two 1x1 launches is a K loop on 132-264 tiles that
    nothing hides and the 1024-channel block output would be read back by the next conv1.  Bit-identical to the two launches."""

    def __init__(self, name, rows, planes, x, w3, b3, identity, y, w1n, b1n, t1n):
        dev = x.device
        self.name, self.rows, self.planes = name, rows, planes
        prep = lambda w: H.prep_conv_weight(w.to(dev), w.shape[1])[0][:w.shape[0]].contiguous()
        fb = lambda b: b.float().to(dev).contiguous()
        self.x, self.identity, self.y, self.t1n = x, identity, y, t1n
        self.w3, self.b3, self.w1n, self.b1n = prep(w3), fb(b3), prep(w1n), fb(b1n)
        self.x_block = None
        self.flops = 2.0 * rows * planes * 4 * planes * 2
        self.bytes = rows * planes * 2 * 2 + 2 * rows * 4 * planes * 2 + (self.w3.numel() + self.w1n.numel()) * 2
        self.convs_in_launch = 2

    def __call__(self):
        H.conv1x1_pair(self.rows, self.planes, self.x, self.w3, self.b3, self.identity, self.y, self.w1n, self.b1n, self.t1n)


class _GroupedConv(_Conv):
    """G convs of identical shape (own weights, own outputs, shared or own inputs) as ONE launch through the group
    dimension of sm_conv_desc: weights stacked [G][cout_pad][Kp], y = [G * rows, cout], GroupNorm statistics
    [G][batch][nlev][cout/8][2].  Used for the cls / reg tower convs of one depth (sipmask_head.py:252-257)."""

    def __init__(self, eng, name, ws, biases, batch, in_sizes, in_row0, x, x_group_rows, in_cstride, y, y_group_rows,
                 out_row0, out_cstride, flags=0, mode=None):
        G = len(ws)
        self._patch_groups = G
        if mode in ("x3", "x2", "x3p"):
            self._x3_scale = H.x3_weight_scale(ws)              # one accumulator scale per launch: shared by the groups
        _Conv.__init__(self, eng, name, ws[0], biases[0], batch, in_sizes, in_row0, x, in_cstride, 1, 1, y, out_row0,
                       out_cstride, flags=flags, mode=mode)
        dev = eng.device
        if self.x3p:
            prep = lambda w: H.prep_conv_weight_patch_x3p(w.to(dev), self.x3_scale)[0]
        elif self.x3:
            prep = (lambda w: H.prep_conv_weight_patch_x3(w.to(dev), self.x3_scale, terms=self.xterms)[0]) if self.patch else \
                (lambda w: H.prep_conv_weight_x3(w.to(dev), self.x3_scale, self.xterms)[0])
        else:
            prep = (lambda w: H.prep_conv_weight_patch(w.to(dev))[0]) if self.patch else \
                (lambda w: H.prep_conv_weight(w.to(dev), in_cstride)[0])
        packed = [self.w] + [prep(w) for w in ws[1:]]
        assert all(p.shape == packed[0].shape for p in packed)
        self.w = torch.stack(packed).contiguous()
        if biases[0] is not None:
            self.bias = torch.stack([b.float().to(dev) for b in biases]).contiguous()
        d = self.desc
        d.ngroups = G
        d.x_group_rows, d.y_group_rows = x_group_rows, y_group_rows
        d.w_group_stride = packed[0].numel()
        d.bias_group_stride = 0 if biases[0] is None else biases[0].numel()
        d.gn_group_stride = 2 * batch * len(in_sizes) * (ws[0].shape[0] // 8)
        self.flops *= G
        self.mfma_flops *= G
        self.bytes = self.bytes * G - (0 if x_group_rows else (G - 1) * sum(batch * h * ww for h, ww in in_sizes) * in_cstride * 2)


class _LevelConv(_Conv):
    """L 3x3 convs of identical channel shape on L pyramid LEVELS, each level with its OWN weights and bias, as ONE launch
    of the patch-resident kernel (sm_conv_desc.w_level_stride): the FPN's output convs fpn_convs[0..2] on the merged
    laterals (fpn.py:154-157) -- 268 + 68 + 20 tiles at B=4, 800 x 1344, which as three launches are one full round and two
    launch-latency-shaped ones.  `patch` stays False when the patch kernel does not take the shape (the caller then builds
    the separate launches)."""

    def __init__(self, eng, name, ws, biases, batch, sizes, in_row0, x, in_cstride, y, out_row0, out_cstride, flags=0,
                 force=False):
        self._patch_min_work = _LEVEL_CONV_MIN_WORK
        self._patch_force = bool(force)
        _Conv.__init__(self, eng, name, ws[0], biases[0], batch, sizes, in_row0, x, in_cstride, 1, 1, y, out_row0, out_cstride,
                       flags=flags)
        if not self.patch:
            return
        dev = eng.device
        packed = [self.w] + [H.prep_conv_weight_patch(w.to(dev))[0] for w in ws[1:]]
        assert all(p.shape == packed[0].shape for p in packed)
        self.w = torch.stack(packed).contiguous()
        self.desc.w_level_stride = packed[0].numel()
        if biases[0] is not None:
            self.bias = torch.stack([b.float().to(dev) for b in biases]).contiguous()
            self.desc.bias_level_stride = biases[0].numel()
        self.bytes += sum(p.numel() for p in packed[1:]) * 2


class _DeformChoice:
    """FeatureAlign's deformable conv of the x3 head plan as one plan entry with two kernels behind it: `window`
    (csrc/deform_patch_x3.hip: the f32 window in LDS, GroupNorm statistics fused; None when the shape is not its own) and
    `gather` (conv_f32.hip's gather loader).  The active one is chosen once per plan from the data (SipMaskEngine._tune_deform)
    or adopted from another engine of the same configuration."""

    def __init__(self, window, gather):
        self.window, self.gather = window, gather
        self.active = window if window is not None else gather

    def pick(self, kernel):
        self.active = self.window if (kernel == "window" and self.window is not None) else self.gather

    @property
    def kernel(self):
        return "window" if self.active is self.window else "gather"

    @property
    def fused_stats(self):
        return self.active is self.window and self.window.gn_stats is not None

    def __getattr__(self, name):          # name, mode, flops, mfma_flops, bytes, desc, ... of the active kernel
        # (copy.deepcopy / pickle probe __deepcopy__ / __setstate__ / __reduce_ex__ on an instance whose __init__ has not run)
        if name.startswith("__") or "active" not in self.__dict__:
            raise AttributeError(name)
        return getattr(self.__dict__["active"], name)

    def __setattr__(self, name, value):   # reads are forwarded, so writes to anything but the wrapper's own three fields
        if name in ("window", "gather", "active"):        # would silently land on the wrapper: refuse them
            object.__setattr__(self, name, value)
        else:
            raise AttributeError("_DeformChoice forwards reads only; set %r on .window / .gather" % name)

    def __call__(self):
        self.active()


class MaskRescorer:
    """SipMask++ mask-scoring branch (sipmask_head.py:200-219,635-641) as a launch list over the N = batch*max_num
    cropped probability masks: f32 [N,Hm,Wm] -> bf16 NHWC (1 channel padded to 8) -> six 3x3 stride-2 convs (+bias,
    ReLU) -> 1x1 mask_scoring (+ReLU, f32) -> global max over the last map at the detection's class x box score."""

    def __init__(self, sd, prefix, batch, max_num, hm, wm, device):
        self.device = torch.device(device)
        n = batch * max_num
        self.batch, self.max_num, self.hm, self.wm = batch, max_num, hm, wm
        self.pos_masks = torch.zeros(batch, max_num, hm, wm, dtype=torch.float32, device=self.device)
        self.x0 = torch.empty(n * hm * wm, 8, dtype=BF16, device=self.device)
        self.convs = []
        x, h, w, c = self.x0, hm, wm, 8
        for i in range(6):
            wgt, b = sd[prefix + "convs_scoring.%d.conv.weight" % i], sd[prefix + "convs_scoring.%d.conv.bias" % i]
            oh, ow = _conv_out(h, 3, 2, 0), _conv_out(w, 3, 2, 0)
            if min(oh, ow) < 1:
                raise ValueError("mask grid %dx%d is too small for the six stride-2 scoring convs" % (hm, wm))
            co = wgt.shape[0]
            y = torch.empty(n * oh * ow, co, dtype=BF16, device=self.device)
            self.convs.append(_Conv(self, "rescore.convs_scoring.%d" % i, wgt, b, n, [(h, w)], [0], x, c, 2, 0, y, [0], co,
                                    flags=SM_CONV_RELU, cin_pad=8 if i == 0 else None))
            x, h, w, c = y, oh, ow, co
        wgt, b = sd[prefix + "mask_scoring.weight"], sd[prefix + "mask_scoring.bias"]
        self.hw, self.ncls = h * w, wgt.shape[0]
        self.feat = torch.empty(n * h * w, self.ncls, dtype=torch.float32, device=self.device)
        self.convs.append(_Conv(self, "rescore.mask_scoring", wgt, b, n, [(h, w)], [0], x, c, 1, 0, self.feat, [0],
                                self.ncls, flags=SM_CONV_RELU | SM_CONV_OUT_F32))
        self.scores = torch.zeros(batch, max_num, dtype=torch.float32, device=self.device)

    def run(self, labels, det, ndet):
        n = self.batch * self.max_num
        H.nchw_to_nhwc_bf16(self.pos_masks.view(n, 1, self.hm, self.wm), self.x0, 8)
        for c in self.convs:
            c()
        return H.mask_rescore(self.feat, labels, det, ndet, self.hw, self.scores)


class SipMaskEngine:
    """Static launch plan for SipMask-R50/R101 inference at a fixed (batch, H, W)."""

    def __init__(self, state_dict, batch, img_hw, depth=50, test_cfg=None, num_classes=81, device="cuda",
                 strides=(8, 16, 32, 64, 128), img_shape=None, head_sizes=None, ssd_flag=False, scale_factor=1.0,
                 rescale=False, vis=False, benchmark=None, precision="bf16", sub_plan=False, scale_factor_max=None,
                 pipelined=False):
        _lib.load()   # fail loudly before anything else if the HIP library is missing
        # sub_plan: this engine is one chain of a SubBatchPlan -- the other chain fills the CUs a short launch leaves
        # idle, so split-K (an extra reduce launch to fill them) only costs: 966 vs 962 img/s (profiles/r02e_ab_subplans.json)
        self.split_k = _SPLIT_K and not sub_plan and not pipelined
        # pipelined: this engine is one slot of a PipelinedPlan -- the steps in flight next to it fill the CUs a launch leaves
        # idle, so what counts is CU time, not the latency of a launch: no split-K, no shrinking of the implicit-GEMM tiles for
        # occupancy (SM_CONV_DBG_BIG_TILES: fewer operand bytes per FLOP; identical results), no side lanes.  Measured at three
        # steps in flight: 1 263 vs 1 211 img/s (big tiles +2.7 %, no split-K +0.9 %); inside the lock-stepped chains of ONE step
        # the same flag LOSES 5 % (round 2).
        self.extra_conv_flags = _lib.SM_CONV_DBG_BIG_TILES if pipelined else 0
        # ... and for the same reason its patch convs keep the uniform 256-position launch: the mixed launch (256-position
        # tiles + 128/192-position finishing tiles, sm_conv3x3_patch_plan) ends a lone launch 15-20 % sooner but spends
        # 4-6 % more CU time on it, which the other chain would have used (measured 998 vs 991 img/s, profiles/r02g_ab_patch_launch_shape.json)
        self.patch_uniform = (sub_plan and not _PATCH_MIXED_IN_CHAINS) or (pipelined and _PATCH_UNIFORM_IN_SLOTS)
        if precision not in ("bf16", "f32", "head_x3"):
            raise ValueError("precision must be 'bf16' (throughput plan), 'head_x3' (bf16 backbone + FPN, split-precision "
                             "head: the reference head's fp32 arithmetic to ~1e-4 on its logits) or 'f32' (parity plan), "
                             "got %r" % (precision,))
        # "f32": every activation / weight float32, convs on the exact-f32 MFMA kernel (csrc/conv_f32.hip), GroupNorm
        # statistics in double -- the plan that is held to the fp32 reference within accumulation-order rounding
        self.precision = precision
        self.act_dtype = torch.float32 if precision == "f32" else BF16
        if not torch.cuda.is_available():
            raise RuntimeError("SipMaskEngine needs a HIP device")
        self.device = torch.device(device)
        self.batch = batch
        self.H, self.W = img_hw
        assert self.H % 32 == 0 and self.W % 32 == 0, "pad images to a multiple of 32 (Pad size_divisor=32)"
        self.depth = depth
        self.ncls = num_classes - 1
        self.strides = tuple(strides)
        self.cfg = dict(nms_pre=1000, score_thr=0.05, nms=dict(type="nms", iou_thr=0.5), max_per_img=100)
        if test_cfg:
            self.cfg.update(test_cfg)
        self.img_shape = img_shape or (self.H, self.W, 3)
        # ssd_flag configs: fast_nms + per-axis mask upsampling (sipmask_head.py:594-605,629-630)
        self.ssd_flag, self.scale_factor, self.rescale = bool(ssd_flag), scale_factor, rescale
        # img_shape / scale_factor are the DEFAULT of every image; set_image_metas() gives each image its own before a
        # run (sipmask_head.py:517-541).  The mask canvas is sized for `scale_factor` (pass the SMALLEST the plan will
        # see: masks are upsampled by 2 / scale_factor), the kernels' source windows for `scale_factor_max`.
        self.scale_factor_max = scale_factor if scale_factor_max is None else scale_factor_max
        # SipMask-VIS head (V/mmdet/models/anchor_heads/sipmask_head.py): track branch, always fast_nms with
        # cfg.max_per_img, mask threshold 0.5, crop/upsample scaled only when rescale (:734-764)
        self.vis = bool(vis)
        self.mask_thr = 0.5 if self.vis else 0.4
        # maskrcnn-benchmark variant (B/ = SipMask-benchmark/): dict(pre_nms_thresh, pre_nms_top_n, nms_thresh,
        # post_top_n) -> relu(scale(bbox_pred)), (location, class)-pair candidates, same-label NMS
        self.benchmark = dict(benchmark) if benchmark else None
        self.steps = []        # (label, callable)
        self.lanes = []        # per step: 0 (main stream), n > 0 (side stream n) or ("join", n...)
        self._side_streams = {}
        # independent branches (bottleneck downsample, FPN output convs of the coarse levels) on side streams: their
        # launches are 20-130 blocks, far below the 512 resident blocks of the chip
        self.multi_stream = os.environ.get("SIPMASK_MULTI_STREAM", "1") != "0"
        self.convs = []        # _Conv objects (for FLOP accounting / per-kernel timing)
        self.fused = []        # _BottleneckTail launches (several convs each; counted in total_conv_flops)
        self.head_start = 0
        if head_sizes is None:
            self._build(state_dict)
        else:                  # head-only plan on caller-provided FPN features
            sd = {k: v.detach() for k, v in state_dict.items()}
            self.lv = H.Levels(batch, head_sizes)
            # (the x3 head splits f32 features itself: a head-only plan takes the caller's f32 features as they are)
            self.pyr = self._buf(self.lv.rows, 256, torch.float32 if precision == "head_x3" else None)
            self._build_head(sd)
            self._build_post()

    @classmethod
    def for_head(cls, state_dict, batch, sizes, num_classes=81, strides=(8, 16, 32, 64, 128), test_cfg=None,
                 img_shape=None, ssd_flag=False, vis=False, benchmark=None, precision="bf16", pipelined=False):
        """Plan for SipMaskHead.forward / get_bboxes alone (features come from the caller).  pipelined: built like a slot
        of a PipelinedPlan (big tiles, no split-K, no side lanes) -- the head of the plan bench.py times."""
        h0, w0 = sizes[0]
        img_hw = (h0 * strides[0], w0 * strides[0])
        eng = cls(state_dict, batch, img_hw, 50, test_cfg, num_classes, "cuda", strides,
                  img_shape or (img_hw[0], img_hw[1], 3), head_sizes=list(sizes), ssd_flag=ssd_flag, vis=vis, benchmark=benchmark,
                  precision=precision, pipelined=pipelined)
        if pipelined:
            eng.multi_stream = False
        return eng

    def load_pyramid(self, feats):
        """copy caller features (tuple of NCHW float tensors) into the bf16 pyramid tensor"""
        lv = self.lv
        for l, f in enumerate(feats):
            h, w = lv.sizes[l]
            assert tuple(f.shape) == (self.batch, 256, h, w), (tuple(f.shape), (self.batch, 256, h, w))
            to_rows = H.nchw_to_nhwc_f32 if self.pyr.dtype == torch.float32 else H.nchw_to_nhwc_bf16
            to_rows(f.detach().float().contiguous(), self.pyr[lv.row0[l]:lv.row0[l] + self.batch * h * w], 256)

    POST_STEPS = ("det_select", "nms", "mask_assemble", "track_gather", "rescore")

    def run_convs(self, img):
        """extract_feat + bbox_head only (SingleStageDetector.forward_dummy, single_stage.py:52-59: the conv-only entry
        `tools/get_flops.py` uses): every launch of the plan except post-processing.  Returns head_outputs()."""
        assert img.shape == (self.batch, 3, self.H, self.W) and img.dtype == torch.float32 and img.is_cuda
        self.img = img.contiguous()
        sel = [i for i in range(len(self.steps)) if self.steps[i][0] not in self.POST_STEPS]
        self._run_steps([self.steps[i] for i in sel], [self.lanes[i] for i in sel])
        return self.head_outputs()

    def run_head(self, with_post=False):
        post = ("det_select", "nms", "mask_assemble", "track_gather", "rescore")
        sel = [i for i in range(self.head_start, len(self.steps)) if with_post or self.steps[i][0] not in post]
        self._run_steps([self.steps[i] for i in sel], [self.lanes[i] for i in sel])

    # -------------------------------------------------------------------------------- helpers
    def _buf(self, rows, c, dtype=None):
        """activation buffer; dtype None = the plan's activation type (bf16, or f32 in the parity plan)"""
        return torch.empty(rows, c, dtype=dtype or self.act_dtype, device=self.device)

    def _add(self, label, fn, lane=0):
        """lane 0 = the caller's stream; lanes > 0 are side HIP streams for branches that do not depend on what
        lane 0 does next (they fork at their first step and are joined by an explicit _join)."""
        self.steps.append((label, fn))
        self.lanes.append(lane)

    def _add_conv(self, conv, lane=0):
        self.convs.append(conv)
        self._add("conv:" + conv.name, conv, lane)
        return conv

    def _join(self, *lanes):
        """lane 0 waits for the side lanes (their results are read by the following steps)"""
        self.steps.append(("join", lambda: None))
        self.lanes.append(("join",) + tuple(lanes))

    def _run_steps(self, steps, lanes):
        """Launch the plan.  Side lanes are torch streams: a lane forks (waits for everything lane 0 has queued so
        far) at its first step after a join, so a step on a side lane may read anything produced before it in plan
        order on lane 0; capture-safe (the side streams join the capture through wait_stream)."""
        # diagnostic (tools/marginal_cost.sh): launches whose label matches SIPMASK_DIAG_SKIP are left out of CAPTURED graphs
        # only -- the eager run before the capture has filled every buffer, so what follows a skipped launch reads plausible
        # (stale) data and the replay's time is the step's time WITHOUT those launches: the marginal cost of a stage inside
        # the pipelined step, which per-launch timings cannot give (the steps in flight overlap)
        skip = _DIAG_SKIP if (_DIAG_SKIP is not None and torch.cuda.is_current_stream_capturing()) else None
        if skip is not None:
            kept = [i for i, (label, _) in enumerate(steps) if label == "join" or not skip.search(label)]
            steps, lanes = [steps[i] for i in kept], [lanes[i] for i in kept]
        if not self.multi_stream:
            for _, fn in steps:
                fn()
            return
        main = torch.cuda.current_stream()
        active = set()
        for (label, fn), lane in zip(steps, lanes):
            if isinstance(lane, tuple):                      # join
                for ln in lane[1:]:
                    if ln in active:
                        main.wait_stream(self._side(ln))
                        active.discard(ln)
                continue
            if lane == 0:
                fn()
                continue
            side = self._side(lane)
            if lane not in active:
                side.wait_stream(main)
                active.add(lane)
            with torch.cuda.stream(side):
                fn()
        for ln in list(active):                              # never leave a lane dangling
            main.wait_stream(self._side(ln))

    def _side(self, lane):
        st = self._side_streams.get(lane)
        if st is None:
            st = self._side_streams[lane] = torch.cuda.Stream(device=self.device)
        return st

    # -------------------------------------------------------------------------------- plan
    def _build(self, sd):
        B, Himg, Wimg, dev = self.batch, self.H, self.W, self.device
        sd = {k: v.detach() for k, v in sd.items()}
        # ---- stem
        f32 = self.precision == "f32"
        cpad = 4 if f32 else 8
        h1, w1 = _conv_out(Himg, 7, 2, 3), _conv_out(Wimg, 7, 2, 3)
        w, b = fold_bn(sd["backbone.conv1.weight"], sd, "backbone.bn1")
        h2, w2 = _conv_out(h1, 3, 2, 1), _conv_out(w1, 3, 2, 1)
        x = self._buf(B * h2 * w2, 64)
        if _STEM_FUSED and not f32:
            # conv1 + bn1 + relu + maxpool (resnet.py:497-505) as ONE launch from the NCHW f32 image (csrc/stem_fused.hip):
            # the 400 x 672 x 64 conv output never goes to HBM
            self.stem_w = H.prep_stem_weight(w.to(dev))
            self.stem_b = b.float().to(dev).contiguous()
            self.stem_flops = 2.0 * B * h1 * w1 * 64 * 147
            self._add("stem_fused", lambda y=x: H.stem_fused(self.img, self.stem_w, self.stem_b, y))
        else:
            self.img_nhwc = self._buf(B * Himg * Wimg, cpad)
            stem = self._buf(B * h1 * w1, 64)
            to_rows = H.nchw_to_nhwc_f32 if f32 else H.nchw_to_nhwc_bf16
            self._add("nhwc", lambda: to_rows(self.img, self.img_nhwc, cpad))
            self._add_conv(_Conv(self, "stem", w, b, B, [(Himg, Wimg)], [0], self.img_nhwc, cpad, 2, 3, stem, [0], 64,
                                 flags=SM_CONV_RELU, cin_pad=cpad))
            pool = H.maxpool3x3s2_f32 if f32 else H.maxpool3x3s2
            self._add("maxpool", (lambda s=stem, y=x: pool(s, y, B, h1, w1, 64)))
        # ---- residual stages (caffe style: stride on conv1, resnet.py:125-130)
        cur, ch, cw, cc = x, h2, w2, 64
        # A/B (SIPMASK_LATENCY_1X1 = "256", "256,512" ...): the 1x1 convs of those stages keep the latency-shaped plan inside
        # pipelined slots (no big-tile flag: 64 x 64 tiles, four blocks per CU) like lat2 / P6 / P7
        lat_planes = [int(v) for v in _LATENCY_1X1]
        lat_1x1 = lambda pl: (True if (getattr(self, "extra_conv_flags", 0) and pl in lat_planes) else None)
        feats = []
        chained_t1 = None
        for li, nblocks in enumerate(ARCH[self.depth]):
            planes = 64 * 2 ** li
            for bi in range(nblocks):
                p = "backbone.layer%d.%d" % (li + 1, bi)
                s = 2 if (bi == 0 and li > 0) else 1
                oh, ow = _conv_out(ch, 1, s, 0), _conv_out(cw, 1, s, 0)
                has_dcn = (p + ".conv2.conv_offset.weight") in sd
                fuse = _FUSE_BOTTLENECK if (not f32 and planes in (64, 128) and not has_dcn) else 0
                # layer1's first block: the 1x1 shortcut conv (64 -> 256, stride 1) rides the fused tail's conv3 (round 4)
                # (... and layer2's: 256 -> 512 at stride 2, SIPMASK_FUSE_SHORTCUT=2)
                fuse_ds = bool(fuse) and bi == 0 and ((_FUSE_SHORTCUT >= 1 and s == 1 and planes == 64 and cc == 64) or
                                                      (_FUSE_SHORTCUT >= 2 and s == 2 and planes == 128 and cc == 256))
                shortcut = None
                if bi == 0 and fuse_ds:
                    wd, bd = fold_bn(sd[p + ".downsample.0.weight"], sd, p + ".downsample.1")
                    shortcut, idt = (wd, bd, cur, s, (ch, cw)), None
                elif bi == 0:     # the shortcut conv reads the block input only: it forks onto a side lane BEFORE
                    # conv1/conv2 are queued and is joined right before conv3 adds it
                    wd, bd = fold_bn(sd[p + ".downsample.0.weight"], sd, p + ".downsample.1")
                    idt = self._buf(B * oh * ow, planes * 4)
                    self._add_conv(_Conv(self, p + ".downsample", wd, bd, B, [(ch, cw)], [0], cur, cc, s, 0, idt, [0],
                                         planes * 4), lane=1)
                else:
                    idt = cur
                wa, ba = fold_bn(sd[p + ".conv1.weight"], sd, p + ".bn1")
                if chained_t1 is not None:       # the previous block's fused launch already produced this conv1
                    t1, chained_t1 = chained_t1, None
                else:
                    t1 = self._buf(B * oh * ow, planes)
                    self._add_conv(_Conv(self, p + ".conv1", wa, ba, B, [(ch, cw)], [0], cur, cc, s, 0, t1, [0], planes,
                                         flags=SM_CONV_RELU, split_k=lat_1x1(planes)))
                wb, bb = fold_bn(sd[p + ".conv2.weight"], sd, p + ".bn2")
                wc, bc = fold_bn(sd[p + ".conv3.weight"], sd, p + ".bn3")
                out = self._buf(B * oh * ow, planes * 4)
                if fuse:
                    if bi == 0 and shortcut is None:
                        self._join(1)
                    next1 = None
                    pn = "backbone.layer%d.%d" % (li + 1, bi + 1)
                    # the next block's conv1 chained behind this launch: everywhere with SIPMASK_FUSE_BOTTLENECK=2 (the round-2
                    # A/B), in layer1 by default (round 4: 8 KB weight slices took the chained kernel from 168 VGPRs + spills
                    # to three clean blocks per CU; the 256-channel block output is then read once instead of twice)
                    if ((fuse >= 2 or (_CHAIN_CONV1 and (planes == 64 or _CHAIN_CONV1 == 3)
                                       and not (_CHAIN_CONV1 == 2 and shortcut is not None)))
                            and not (shortcut is not None and planes != 64)       # (the stride-2 shortcut kernel has no chain)
                            and bi + 1 < nblocks
                            and (pn + ".conv2.conv_offset.weight") not in sd):
                        wn, bn = fold_bn(sd[pn + ".conv1.weight"], sd, pn + ".bn1")
                        chained_t1 = self._buf(B * oh * ow, planes)
                        next1 = (wn, bn, chained_t1)
                    tail = _BottleneckTail(p + ".tail", B, (oh, ow), planes, t1, wb, bb, wc, bc, idt, out, next1,
                                           shortcut=shortcut)
                    self.fused.append(tail)
                    self._add("conv:" + tail.name, tail)
                    cur, ch, cw, cc = out, oh, ow, planes * 4
                    continue
                t2 = self._buf(B * oh * ow, planes)
                if has_dcn:
                    # SipMask++ backbone DCN (DeformConvPack, deform_conv.py:258-296): offsets from an ordinary
                    # 3x3 conv (f32), then the deformable conv with bn2 folded in (it is linear in the weight)
                    w_off, b_off = sd[p + ".conv2.conv_offset.weight"], sd[p + ".conv2.conv_offset.bias"]
                    dg = w_off.shape[0] // 18
                    off = self._buf(B * oh * ow, 18 * dg, torch.float32)
                    self._add_conv(_Conv(self, p + ".conv2.conv_offset", w_off, b_off, B, [(oh, ow)], [0], t1, planes, 1,
                                         1, off, [0], 18 * dg, flags=SM_CONV_OUT_F32))
                    self._add_conv(_Conv(self, p + ".conv2", wb, bb, B, [(oh, ow)], [0], t1, planes, 1, 1, t2, [0],
                                         planes, flags=SM_CONV_RELU, deform_groups=dg, offset=off))
                else:
                    self._add_conv(_Conv(self, p + ".conv2", wb, bb, B, [(oh, ow)], [0], t1, planes, 1, 1, t2, [0],
                                         planes, flags=SM_CONV_RELU))
                if bi == 0:
                    self._join(1)
                pn = "backbone.layer%d.%d" % (li + 1, bi + 1)
                if (_PAIR_1X1 and not f32 and planes == 256 and bi + 1 < nblocks
                        and (pn + ".conv2.conv_offset.weight") not in sd):
                    # layer3: conv3 of this block + conv1 of the next as one launch (round 4)
                    wn, bn = fold_bn(sd[pn + ".conv1.weight"], sd, pn + ".bn1")
                    chained_t1 = self._buf(B * oh * ow, planes)
                    pair = _Conv1x1Pair(p + ".conv3+", B * oh * ow, planes, t2, wc, bc, idt, out, wn, bn, chained_t1)
                    self.fused.append(pair)
                    self._add("conv:" + pair.name, pair)
                else:
                    self._add_conv(_Conv(self, p + ".conv3", wc, bc, B, [(oh, ow)], [0], t2, planes, 1, 0, out, [0],
                                         planes * 4, flags=SM_CONV_RELU | SM_CONV_RES_ADD, residual=idt,
                                         res_cstride=planes * 4, split_k=lat_1x1(planes)))
                cur, ch, cw, cc = out, oh, ow, planes * 4
            feats.append((cur, ch, cw, cc))
        self.backbone_feats = feats
        # ---- FPN (start_level=1): laterals with fused top-down nearest add (fpn.py:141-152)
        sizes = [(f[1], f[2]) for f in feats[1:]]
        p6 = (_conv_out(sizes[2][0], 3, 2, 1), _conv_out(sizes[2][1], 3, 2, 1))
        p7 = (_conv_out(p6[0], 3, 2, 1), _conv_out(p6[1], 3, 2, 1))
        self.lv = H.Levels(B, sizes + [p6, p7])
        lv = self.lv
        lats = [None] * 3
        self.pyr = self._buf(lv.rows, 256)
        # the three merged laterals live in ONE row tensor (levels 0-2 of a pyramid layout), so that the three output convs
        # can read them as the levels of one launch
        lv3 = H.Levels(B, sizes)
        lat_rows = self._buf(lv3.rows, 256)
        for i in range(3):
            lats[i] = lat_rows[lv3.row0[i]:lv3.row0[i] + B * sizes[i][0] * sizes[i][1]]

        def out_conv(i, lane):
            self._add_conv(_Conv(self, "fpn.out%d" % i, sd["neck.fpn_convs.%d.conv.weight" % i],
                                 sd["neck.fpn_convs.%d.conv.bias" % i], B, [sizes[i]], [0], lats[i], 256, 1, 1,
                                 self.pyr, [lv.row0[i]], 256), lane)

        # fpn_convs[0..2] as ONE launch with per-level weights (round 4; _LevelConv): at B=4, 800 x 1344 the three launches
        # were 268 / 68 / 20 tiles -- one full round and two launch-latency-shaped ones (0.106 + 0.045 + 0.043 ms)
        small_split = True if getattr(self, "extra_conv_flags", 0) else None    # pipelined slot: see _Conv(split_k=...)
        grouped = None
        if self.precision != "f32" and _FPN_GROUPED != "0":
            g = _LevelConv(self, "fpn.outs", [sd["neck.fpn_convs.%d.conv.weight" % i] for i in range(3)],
                           [sd["neck.fpn_convs.%d.conv.bias" % i] for i in range(3)], B, sizes, list(lv3.row0), lat_rows, 256,
                           self.pyr, list(lv.row0[:3]), 256, force=(_FPN_GROUPED == "1"))
            grouped = g if g.patch else None
        self.fpn_grouped = grouped is not None

        def p6_p7(lane):
            self._add_conv(_Conv(self, "fpn.p6", sd["neck.fpn_convs.3.conv.weight"], sd["neck.fpn_convs.3.conv.bias"],
                                 B, [sizes[2]], [lv.row0[2]], self.pyr, 256, 2, 1, self.pyr, [lv.row0[3]], 256,
                                 split_k=small_split), lane)
            if self.precision == "f32" or not _RELU_COPY_P7:
                self._add_conv(_Conv(self, "fpn.p7", sd["neck.fpn_convs.4.conv.weight"], sd["neck.fpn_convs.4.conv.bias"],
                                     B, [p6], [lv.row0[3]], self.pyr, 256, 2, 1, self.pyr, [lv.row0[4]], 256,
                                     flags=SM_CONV_IN_RELU), lane)
            else:
                # relu(P6) as its own 100-KB tensor: the P7 launch is a handful of tiles with 36 K steps, and the
                # input-ReLU flag would put it on the register-staged loader (one exposed load latency per K step)
                n6 = B * p6[0] * p6[1]
                self.p6_relu = self._buf(n6, 256)
                p6_rows = self.pyr[lv.row0[3]:lv.row0[3] + n6]
                self._add("relu:p6", lambda: H.relu_bf16(p6_rows, self.p6_relu), lane)
                self._add_conv(_Conv(self, "fpn.p7", sd["neck.fpn_convs.4.conv.weight"], sd["neck.fpn_convs.4.conv.bias"],
                                     B, [p6], [0], self.p6_relu, 256, 2, 1, self.pyr, [lv.row0[4]], 256,
                                     split_k=small_split), lane)

        # plan order = dependency order on lane 0 (lat2 -> lat1 -> lat0 -> outputs).  Separate output convs: the coarse ones
        # only need their own lateral, so they leave for side lanes as soon as it is queued: lane 1 = out2 -> P6 -> P7 (66,
        # 20 and 8 blocks), lane 2 = out1 (264 blocks), both overlapped with lat1 / lat0 / out0 on lane 0.  Grouped: the
        # three laterals, the grouped launch, then P6 -> P7 on lane 1 (the head's first convs need all five levels).
        for i in (2, 1, 0):
            f, fh, fw, fc = feats[i + 1]
            wl = sd["neck.lateral_convs.%d.conv.weight" % i]
            bl = sd["neck.lateral_convs.%d.conv.bias" % i]
            if i == 2:
                self._add_conv(_Conv(self, "fpn.lat%d" % i, wl, bl, B, [(fh, fw)], [0], f, fc, 1, 0, lats[i], [0], 256,
                                     split_k=small_split))
                if grouped is None:
                    out_conv(2, 1)
                    p6_p7(1)
            else:
                self._add_conv(_Conv(self, "fpn.lat%d" % i, wl, bl, B, [(fh, fw)], [0], f, fc, 1, 0, lats[i], [0], 256,
                                     flags=SM_CONV_RES_NEAREST, residual=lats[i + 1], res_cstride=256,
                                     res_sizes=[sizes[i + 1]], res_row0=[0]))
                if i == 1 and grouped is None:
                    out_conv(1, 2)
        if grouped is None:
            out_conv(0, 0)
        else:
            self._add_conv(grouped)
            p6_p7(1)
        self._join(1, 2)
        self.head_start = len(self.steps)
        self._build_head(sd)
        self._build_post()

    def _gn(self, label, x, gamma, beta, conv=None, lane=0, stats=None):
        """GroupNorm(32)+ReLU in place.  With ``conv`` (the launch that produced x) the statistics pass is
        fused into that conv's epilogue and only the normalisation kernel remains.  Concurrent lanes need their own
        statistics buffer."""
        g = gamma.float().to(self.device).contiguous()
        b = beta.float().to(self.device).contiguous()
        st = self.gn_stats if stats is None else stats
        if self.precision == "f32":          # statistics pass in double + normalise pass (no fused epilogue statistics)
            st64 = self.gn_stats64 if stats is None else self._gn_stats64_for(stats)
            self._add("gn:" + label, lambda: H.groupnorm_f32(x, x, g, b, st64, self.lv, 256, 32, 1e-5, True), lane)
            return
        if conv is not None:
            conv.gn_stats = st
            self._add("gn:" + label, lambda: H.groupnorm_apply(x, x, g, b, st, self.lv, 256, 32, 1e-5, True), lane)
        else:
            self._add("gn:" + label, lambda: H.groupnorm(x, x, g, b, st, self.lv, 256, 32, 1e-5, True), lane)

    # feat_masks: materialised lazily
    _needs_basis = False

    def _upsample_basis(self):
        h0, w0 = self._basis_h0w0
        if self._basis is None:
            self._basis = self._buf(self.batch * self.hm * self.wm, 32, torch.float32)
        H.upsample_bilinear(self.basis_lo, self._basis, self.batch, h0, w0, 32, 4, 32, 32, 0, True)
        return self._basis

    def _basis_step(self):
        if self._needs_basis:
            self._upsample_basis()

    @property
    def basis(self):
        """feat_masks as rows [B*Hm*Wm, 32] f32, up to date with the last run"""
        if self._needs_basis and self._basis is not None:
            return self._basis
        return self._upsample_basis()

    def _gn_stats64_for(self, stats):
        """the double-precision twin of a per-lane statistics buffer (f32 plan)"""
        k = stats.data_ptr()
        if k not in self._gn64:
            self._gn64[k] = torch.zeros(stats.numel(), dtype=torch.float64, device=self.device)
        return self._gn64[k]

    def _build_head_x3(self, sd, prefix="bbox_head."):
        """SipMaskHead.forward (sipmask_head.py:241-287) in split precision (`precision="head_x3"`): the reference head is
        fp32, and bf16 operand rounding is what moves the bf16 plan's mask logits by ~1.5.  Here every head activation
        stays f32 between layers; a conv reads its input as two binary16 halves per value, [hi | lo | hi] along the
        channel axis (csrc/split_x3.hip), against weights [hi | hi | lo] on the bf16 plan's own MFMA kernels with the
        f16 instruction (SM_CONV_F16): three half products per element product, f32 accumulation, ~2^-21 per product.
        GroupNorm statistics come out of the conv epilogues as before (fixed point); the normalise pass writes the next
        layer's split operand (and f32 rows where a consumer wants them).  FeatureAlign's deformable conv -- the one
        operand VALU has to produce -- reads f32 rows and splits the blended samples in its loader (csrc/conv_f32.hip, X3)."""
        B, lv, dev, h = self.batch, self.lv, self.device, prefix
        self._sd, self._sd_keys = sd, set(sd.keys())
        sizes, row0, rows = lv.sizes, lv.row0, lv.rows
        f32, F16 = torch.float32, torch.float16
        depth = lambda kind: sum(1 for k in sd if k.startswith(h + kind + "_convs.") and k.endswith(".conv.weight"))
        self.flag_norm = (h + "reg_convs.0.gn.weight") in sd
        if any(k.startswith(h + "track_convs.") for k in sd):
            raise NotImplementedError("precision='head_x3' covers the SipMask head (no VIS track branch)")
        ncls, nreg = depth("cls"), depth("reg")
        if not (1 <= ncls <= nreg):
            raise NotImplementedError("head_x3: cls tower not deeper than the reg tower (every reference config)")
        S = 2 * B * len(lv) * 32
        self.gn_stats = H.gn_stats_alloc(B * len(lv) * 32, dev)
        stats2 = H.gn_stats_alloc(2 * B * len(lv) * 32, dev)
        par = lambda n: sd[n].float().to(dev).contiguous()
        TF = _lib_flag("SM_CONV_DBG_TILE256") | _lib_flag("SM_CONV_DBG_HAND_PLACED")
        # head input: the FPN pyramid as [hi | lo | hi] (bf16 rows of the bf16 backbone, or the caller's f32 features)
        # A bf16 pyramid (the bf16 backbone's) has no low half: [hi | hi] against weights [hi | lo] is the same sum without its
        # zero term -- the first tower launch does 2/3 of the MFMA work (round 5).  f32 features (for_head) keep three terms.
        two = _X3_TOWER0_TWO_TERMS and self.pyr.dtype == torch.bfloat16
        # operand layout of the 3x3 convs behind the first: paired ([hi 16 | lo 16] per 16 channels, 2 * 256 per row: mode "x3p",
        # round 6) or K-concatenated ([hi | lo | hi], 3 * 256 per row: mode "x3")
        pairs = _X3_PAIRS
        self.x3_pairs = pairs
        tw, tmode = (512, "x3p") if pairs else (768, "x3")
        pyr_w = 512 if (two or pairs) else 768
        self.pyr_x3 = torch.empty(rows, pyr_w, dtype=F16, device=dev)
        if two:
            self._add("split:pyr", lambda: H.split2_f16(self.pyr, self.pyr_x3, 256))
        elif pairs:
            self._add("split:pyr", lambda: H.split_pairs_f16(self.pyr, self.pyr_x3, 256))
        else:
            self._add("split:pyr", lambda: H.split3_f16(self.pyr, self.pyr_x3, 256))

        def gn_or_split(label, yv, st, norm_name, keep_f32, out):
            """GroupNorm + ReLU (or nothing: SSD-style towers, whose ReLU is the conv's) of a tower conv's f32 output -> the
            operand its consumers read, in the plan's layout (paired, or [hi | lo | hi]); keep_f32: also f32 rows, in place"""
            key = "y_pairs" if pairs else "y_split"
            if self.flag_norm:
                gam, bet = par(h + norm_name + ".weight"), par(h + norm_name + ".bias")
                self._add("gn:" + label, lambda: H.groupnorm_apply_x3(yv, gam, bet, st, lv, 256, 32, 1e-5, True,
                                                                      y_f32=yv if keep_f32 else None, **{key: out}))
            elif pairs:
                self._add("split:" + label, lambda: H.split_pairs_f16(yv, out, 256))
            else:
                self._add("split:" + label, lambda: H.split3_f16(yv, out, 256))

        relu = 0 if self.flag_norm else SM_CONV_RELU
        x, xg, xw = self.pyr_x3, 0, pyr_w
        cls_f32 = reg_x3 = None
        for i in range(ncls):                         # cls + reg tower convs of one depth = ONE grouped launch
            y = torch.empty(2 * rows, 256, dtype=f32, device=dev)
            names = ["cls_convs.%d" % i, "reg_convs.%d" % i]
            first2 = two and i == 0
            c = self._add_conv(_GroupedConv(self, "head.tower%d" % i, [sd[h + n + ".conv.weight"] for n in names],
                                            [sd.get(h + n + ".conv.bias") for n in names], B, sizes, row0, x, xg,
                                            xw, y, rows, row0, 256, flags=TF | relu,
                                            mode="x2" if first2 else (tmode if (i > 0 or pairs) else "x3")))
            if self.flag_norm:
                c.gn_stats = stats2
            last_depth = i == ncls - 1
            # both groups' next operand in ONE tensor (the next grouped launch reads group g at g * rows); behind the last
            # grouped depth only the reg tower goes on
            nxt = torch.empty(2 * rows, tw, dtype=F16, device=dev) if not last_depth else None
            for g, n in enumerate(names):
                yv, st = y[g * rows:(g + 1) * rows], stats2[g * S:(g + 1) * S]
                last_cls = g == 0 and last_depth                  # feeds FeatureAlign's f32 deformable conv only
                last_reg = g == 1 and i == nreg - 1               # feeds reg_ctr, the mask branch ([hi | lo | hi]) and f32 consumers
                if last_cls:
                    cls_f32 = yv
                    if self.flag_norm:                            # normalise in place, no split operand
                        gam, bet = par(h + n + ".gn.weight"), par(h + n + ".gn.bias")
                        self._add("gn:" + n, (lambda yv=yv, gam=gam, bet=bet, st=st:
                                              H.groupnorm_apply_x3(yv, gam, bet, st, lv, 256, 32, 1e-5, True, y_f32=yv)))
                    continue
                # the operand of what reads this tensor: the next tower conv (both groups in `nxt`), or -- behind the reg tower's
                # last conv -- reg_ctr and sip_mask_lat0's convs, which also read the f32 rows
                out = nxt[g * rows:(g + 1) * rows] if nxt is not None else torch.empty(rows, tw, dtype=F16, device=dev)
                gn_or_split(n, yv, st, n + ".gn", last_reg, out)
                if last_reg:
                    self.reg_feat, reg_x3 = yv, out
                else:
                    xr = out
            x, xg, xw = nxt, rows, tw
        for i in range(ncls, nreg):                   # the reg tower is deeper (stacked_convs vs stacked_convs - 1)
            y = torch.empty(rows, 256, dtype=f32, device=dev)
            name = "reg_convs.%d" % i
            c = self._add_conv(_Conv(self, "head." + name, sd[h + name + ".conv.weight"], sd.get(h + name + ".conv.bias"), B,
                                     sizes, row0, xr, tw, 1, 1, y, row0, 256, flags=relu, mode=tmode))
            if self.flag_norm:
                c.gn_stats = self.gn_stats
            last = i == nreg - 1
            out = torch.empty(rows, tw, dtype=F16, device=dev)       # the next reg conv's operand; behind the last: reg_ctr's / sip_mask_lat0's
            gn_or_split(name, y, self.gn_stats, name + ".gn", last, out)
            xr = out
            if last:
                self.reg_feat, reg_x3 = y, out
        self.cls_feat = cls_f32
        # mask basis branch (sipmask_head.py:275-285) on lane 2: f32 [l0 | up2(l1) | up4(l2)] -> split -> 1x1 -> split -> 3x3
        (h0, w0) = sizes[0]
        n0 = B * h0 * w0
        # [l0 | up2(l1) | up4(l2)] written straight as the split operand of sip_mask_lat0 (768 channels -> 3 x 768 halves),
        # and that 1x1 conv writes ITS output as the split operand of sip_mask_lat (SM_CONV_OUT_X3): no f32 round trips
        w_l0 = sd[h + "sip_mask_lat0.weight"]
        exact_grids = all(sizes[l][0] * 2 ** l == h0 and sizes[l][1] * 2 ** l == w0 for l in range(3)) and h0 % 4 == 0 and w0 % 4 == 0
        self.lat0_by_linearity = _LAT0_LINEAR and tuple(w_l0.shape) == (512, 768, 1, 1) and exact_grids
        # sip_mask_lat's operand: paired like the towers' when the branch runs by linearity (round 6: the 1x1 convs on the
        # implicit-GEMM kernel's paired 32-wide-K loop, sm_upsample_sum2 writing pairs, sip_mask_lat on the small-cout kernel's
        # paired instantiation); the concatenating fallback keeps [hi | lo | hi] end to end
        lat_pairs = pairs and self.lat0_by_linearity
        lw, lmode = (2, "x3p") if lat_pairs else (3, "x3")
        lat0_x3 = torch.empty(n0, lw * 512, dtype=F16, device=dev)
        if self.lat0_by_linearity:
            # sip_mask_lat0 by linearity (see _build_head): three x3 convs on the levels' split operands (reg_x3 holds all
            # levels), f32 outputs; sm_upsample_sum2 adds them on the fine grid in f32, applies the ReLU and writes the split
            # operand of sip_mask_lat -- the same arithmetic as upsample-then-conv up to f32 rounding.
            outs = [torch.empty(B * sizes[l][0] * sizes[l][1], 512, dtype=f32, device=dev) for l in range(3)]
            for l in (1, 2, 0):
                self._add_conv(_Conv(self, "head.sip_mask_lat0" + ("" if l == 0 else ".l%d" % l),
                                     w_l0[:, 256 * l:256 * (l + 1)].contiguous(), sd[h + "sip_mask_lat0.bias"] if l == 0 else None,
                                     B, [sizes[l]], [row0[l]], reg_x3, tw, 1, 0, outs[l], [0], 512, mode=tmode), 2)
            self._add("up:sum2", lambda: H.upsample_sum2(outs[0], outs[1], outs[2], lat0_x3, B, h0, w0, 512, relu=True), 2)
        else:
            cat_x3 = torch.empty(n0, 3 * 768, dtype=F16, device=dev)
            for l in range(3):
                fh, fw = sizes[l]
                src = self.reg_feat[row0[l]:row0[l] + B * fh * fw]
                self._add("up:cat%d" % l, (lambda s=src, fh=fh, fw=fw, l=l: H.upsample_bilinear_x3(
                    s, cat_x3, B, fh, fw, 256, 2 ** l, 768, 256 * l)), 2)
            self._add_conv(_Conv(self, "head.sip_mask_lat0", w_l0, sd[h + "sip_mask_lat0.bias"],
                                 B, [(h0, w0)], [0], cat_x3, 3 * 768, 1, 0, lat0_x3, [0], 3 * 512,
                                 flags=SM_CONV_RELU | _lib.SM_CONV_OUT_X3, mode="x3"), 2)
        self.basis_lo = torch.empty(n0, 32, dtype=f32, device=dev)
        self._add_conv(_Conv(self, "head.sip_mask_lat", sd[h + "sip_mask_lat.weight"], sd[h + "sip_mask_lat.bias"], B,
                             [(h0, w0)], [0], lat0_x3, lw * 512, 1, 1, self.basis_lo, [0], 32, flags=SM_CONV_RELU, mode=lmode), 2)
        self.hm, self.wm = 4 * h0, 4 * w0
        self._basis = None
        self._basis_h0w0 = (h0, w0)
        self._add("up:basis", lambda: self._basis_step(), 2)
        # fcos_reg (4, x Scale) + fcos_centerness (1): one 5-channel conv on the reg tower's split output
        # (+ 3 zero channels: 8 couts = one 16-byte store per position and the patch kernel's 32-cout tile)
        w_rc = torch.cat([sd[h + "fcos_reg.weight"], sd[h + "fcos_centerness.weight"],
                          torch.zeros_like(sd[h + "fcos_reg.weight"][:3])], 0)
        b_rc = torch.cat([sd[h + "fcos_reg.bias"], sd[h + "fcos_centerness.bias"], torch.zeros_like(sd[h + "fcos_reg.bias"][:3])], 0)
        scales = [float(sd[h + "scales.%d.scale" % i]) for i in range(len(lv))]
        self.reg_out = torch.zeros(rows, 8, dtype=f32, device=dev)
        c = self._add_conv(_Conv(self, "head.reg_ctr", w_rc, b_rc, B, sizes, row0, reg_x3, tw, 1, 1, self.reg_out, row0, 8,
                                 flags=(_lib.SM_CONV_RELU_NCH if self.benchmark else 0), scale_nch=4, level_scale=scales,
                                 mode=tmode))
        c.flops, c.mfma_flops = c.flops * 5 / 8, c.mfma_flops * 5 / 8      # the 3 zero channels are not work (FLOP accounting)
        # FeatureAlign: offsets (f32 1x1 of the box prediction) -> deformable conv in exact f32 -> GN + ReLU -> split
        self.w_off = sd[h + "feat_align.conv_offset.weight"].float().view(72, 4).to(dev).contiguous()
        self.offsets = torch.empty(rows, 72, dtype=f32, device=dev)
        self._add("offset", lambda: H.offset_linear(self.reg_out, 8, self.w_off, lv, self.offsets))
        self.aligned = torch.empty(rows, 256, dtype=f32, device=dev)
        fa_args = (self, "head.feat_align", sd[h + "feat_align.conv_adaption.weight"],
                   sd.get(h + "feat_align.conv_adaption.bias"), B, sizes, row0, self.cls_feat, 256, 1, 1, self.aligned, row0, 256)
        fa_kw = dict(deform_groups=4, offset=self.offsets, flags=(0 if self.flag_norm else SM_CONV_RELU))
        # two kernels can run this conv: the LDS-window kernel (csrc/deform_patch_x3.hip, round 5: 0.80 -> 0.3 ms per four
        # images; its epilogue also produces the GroupNorm statistics) and conv_f32.hip's gather loader, which costs the same
        # whatever the offsets are.  SIPMASK_X3_FEAT_ALIGN = "window" (default: the window kernel unless the checkpoint's
        # offsets mostly leave its window -- the bf16 plan's rule, _tune_deform), "f32x3" / "f32" pin the gather kernels.
        gather = _Conv(*fa_args, mode=("f32" if _X3_FEAT_ALIGN == "f32" else "f32x3"), **fa_kw)
        window = None
        if _X3_FEAT_ALIGN not in ("f32", "f32x3"):
            dq = H.make_conv_desc(B, sizes, sizes, row0, row0, 256, 256, 256, 3, 1, 1, 256, 256, deform_groups=4,
                                  flags=SM_CONV_OUT_F32 | _lib.SM_CONV_F16)
            if H.deform_conv2d_x3_supported(dq):
                window = _Conv(*fa_args, mode="x3w", **fa_kw)
                if self.flag_norm:
                    window.gn_stats = self.gn_stats
        fa = self._fa_conv = _DeformChoice(window, gather)
        self._add_conv(fa)
        self._deform_tune = window is not None and _DEFORM_MODE == "auto"
        if window is not None and _DEFORM_MODE == "1":
            fa.pick("gather")
        self.deform_choice = None
        aligned_x3 = torch.empty(rows, tw, dtype=F16, device=dev)
        if self.flag_norm:
            gam, bet = par(h + "feat_align.norm.weight"), par(h + "feat_align.norm.bias")
            # (the window kernel's epilogue has written the statistics already)
            self._add("gn_stats:feat_align", lambda: None if fa.fused_stats else H.gn_stats_f32_fix(self.aligned, self.gn_stats, lv, 256, 32))
            self._add("gn:feat_align", lambda: H.groupnorm_apply_x3(self.aligned, gam, bet, self.gn_stats, lv, 256, 32, 1e-5,
                                                                    True, **{"y_pairs" if pairs else "y_split": aligned_x3}))
        elif pairs:
            self._add("split:feat_align", lambda: H.split_pairs_f16(self.aligned, aligned_x3, 256))
        else:
            self._add("split:feat_align", lambda: H.split3_f16(self.aligned, aligned_x3, 256))
        # fcos_cls (80) + sip_cof (128): one 208-channel conv
        w_cc = torch.cat([sd[h + "fcos_cls.weight"], sd[h + "sip_cof.weight"]], 0)
        b_cc = torch.cat([sd[h + "fcos_cls.bias"], sd[h + "sip_cof.bias"]], 0)
        self.ncc = self.ncls + 128
        self.cls_cof = torch.empty(rows, self.ncc, dtype=f32, device=dev)
        self._add_conv(_Conv(self, "head.cls_cof", w_cc, b_cc, B, sizes, row0, aligned_x3, tw, 1, 1, self.cls_cof, row0,
                             self.ncc, mode=tmode))
        self.track_feats = None

    def _build_head(self, sd, prefix="bbox_head."):
        """SipMaskHead.forward, sipmask_head.py:241-287, on the pyramid tensor self.pyr."""
        if self.precision == "head_x3":
            return self._build_head_x3(sd, prefix)
        B, lv, dev, h = self.batch, self.lv, self.device, prefix
        self._sd, self._sd_keys = sd, set(sd.keys())
        sizes, row0 = lv.sizes, lv.row0
        self.gn_stats = H.gn_stats_alloc(B * len(lv) * 32, dev)
        self.gn_stats64 = torch.zeros(B * len(lv) * 32 * 2, dtype=torch.float64, device=dev)
        self._gn64 = {}

        # tower depth / norm as laid down by _init_layers (sipmask_head.py:159-185): stacked_convs-1 cls convs,
        # stacked_convs reg convs; norm_cfg=None (SSD configs) -> conv bias + ReLU, no GroupNorm
        depth = lambda kind: sum(1 for k in sd if k.startswith(h + kind + "_convs.") and k.endswith(".conv.weight"))
        self.flag_norm = (h + "reg_convs.0.gn.weight") in sd

        def tower(kind, n, lane=0, stats=None):
            x = self.pyr
            for i in range(n):
                y = self._buf(lv.rows, 256)
                name = "%s_convs.%d" % (kind, i)
                if self.flag_norm:     # a conv bias in front of GN exists in the B/ variant only (sipmask.py:70-79)
                    c = self._add_conv(_Conv(self, "head." + name, sd[h + name + ".conv.weight"],
                                             sd.get(h + name + ".conv.bias"), B, sizes, row0, x, 256, 1, 1, y, row0, 256),
                                       lane)
                    self._gn(name, y, sd[h + name + ".gn.weight"], sd[h + name + ".gn.bias"], conv=c, lane=lane,
                             stats=stats)
                else:
                    self._add_conv(_Conv(self, "head." + name, sd[h + name + ".conv.weight"],
                                         sd.get(h + name + ".conv.bias"), B, sizes, row0, x, 256, 1, 1, y, row0, 256,
                                         flags=SM_CONV_RELU), lane)
                x = y
            return x

        # the classification tower only meets the box branch at FeatureAlign: it runs on a side lane (own GroupNorm
        # statistics buffer) next to the regression tower; a 1404-block launch leaves the last of its 2.74 rounds
        # of resident blocks 38 % empty, which the other tower's blocks fill
        self.gn_stats_cls = torch.zeros_like(self.gn_stats)
        if _GROUPED_TOWERS and self.precision != "f32" and self.flag_norm and depth("cls") >= 1:
            # cls and reg tower convs of one depth as ONE grouped launch (2 x 353 tiles of 256x256 fill the 256 CUs'
            # rounds to 92 %), each followed by the two towers' GroupNorm passes on two lanes
            n_sh = min(depth("cls"), depth("reg"))
            S = self.gn_stats.numel()
            stats2 = H.gn_stats_alloc(S, dev)       # [2 towers][B][nlev][32][2]
            x, xg = self.pyr, 0
            for i in range(n_sh):
                y = self._buf(2 * lv.rows, 256)
                names = ["cls_convs.%d" % i, "reg_convs.%d" % i]
                c = self._add_conv(_GroupedConv(self, "head.tower%d" % i, [sd[h + n + ".conv.weight"] for n in names],
                                                [sd.get(h + n + ".conv.bias") for n in names], B, sizes, row0, x, xg, 256, y,
                                                lv.rows, row0, 256,
                                                flags=_lib_flag("SM_CONV_DBG_TILE256") | _lib_flag("SM_CONV_DBG_HAND_PLACED")))
                c.gn_stats = stats2
                for g, n in enumerate(names):
                    yv, st = y[g * lv.rows:(g + 1) * lv.rows], stats2[g * S:(g + 1) * S]
                    gam = sd[h + n + ".gn.weight"].float().to(dev).contiguous()
                    bet = sd[h + n + ".gn.bias"].float().to(dev).contiguous()
                    self._add("gn:" + n, (lambda yv=yv, gam=gam, bet=bet, st=st: H.groupnorm_apply(
                        yv, yv, gam, bet, st, self.lv, 256, 32, 1e-5, True)), 1 if g == 0 else 0)
                self._join(1)
                x, xg = y, lv.rows
            self.cls_feat, x = x[:lv.rows], x[lv.rows:]
            for i in range(n_sh, depth("reg")):                    # the reg tower is one conv deeper
                y = self._buf(lv.rows, 256)
                name = "reg_convs.%d" % i
                c = self._add_conv(_Conv(self, "head." + name, sd[h + name + ".conv.weight"], sd.get(h + name + ".conv.bias"),
                                         B, sizes, row0, x, 256, 1, 1, y, row0, 256))
                self._gn(name, y, sd[h + name + ".gn.weight"], sd[h + name + ".gn.bias"], conv=c)
                x = y
            self.reg_feat = x
            assert depth("cls") == n_sh, "a cls tower deeper than the reg tower does not occur in the reference configs"
        else:
            self.cls_feat = tower("cls", depth("cls"), lane=1, stats=self.gn_stats_cls)
            self.reg_feat = tower("reg", depth("reg"))
        # mask basis branch (sipmask_head.py:275-285): needs reg_feat only and is consumed by mask assembly only, so it
        # runs on lane 2 next to reg_ctr / FeatureAlign / cls_cof and the low-occupancy det_select + NMS (joined in
        # _build_post right before mask assembly)
        (h0, w0) = sizes[0]
        self.lat0 = self._buf(B * h0 * w0, 512)
        w_l0 = sd[h + "sip_mask_lat0.weight"]
        exact_grids = all(sizes[l][0] * 2 ** l == h0 and sizes[l][1] * 2 ** l == w0 for l in range(3)) and h0 % 4 == 0 and w0 % 4 == 0
        self.lat0_by_linearity = (_LAT0_LINEAR and self.precision == "bf16" and tuple(w_l0.shape) == (512, 768, 1, 1) and exact_grids)
        if self.lat0_by_linearity:
            # sip_mask_lat0 by linearity (round 4): a 1x1 conv commutes with bilinear upsampling, so
            #   W . [l0 | up2(l1) | up4(l2)] = W0 . l0 + up2(W1 . l1) + up4(W2 . l2)
            # -- the three products at their own resolutions (23.1 instead of 52.9 GFLOP per 4 images), no 768-channel
            # concatenation (103 MB written and read back); the coarse sum is the RES_ADD residual of the l0 conv, whose
            # epilogue adds the bias and the ReLU as before.
            n1, n2 = B * sizes[1][0] * sizes[1][1], B * sizes[2][0] * sizes[2][1]
            a1, a2 = self._buf(n1, 512), self._buf(n2, 512)
            res = self._buf(B * h0 * w0, 512)
            for l, dst in ((1, a1), (2, a2)):
                self._add_conv(_Conv(self, "head.sip_mask_lat0.l%d" % l, w_l0[:, 256 * l:256 * (l + 1)].contiguous(), None, B,
                                     [sizes[l]], [row0[l]], self.reg_feat, 256, 1, 0, dst, [0], 512), 2)
            self._add("up:sum2", lambda: H.upsample_sum2(None, a1, a2, res, B, h0, w0, 512), 2)
            self._add_conv(_Conv(self, "head.sip_mask_lat0", w_l0[:, :256].contiguous(), sd[h + "sip_mask_lat0.bias"], B,
                                 [(h0, w0)], [row0[0]], self.reg_feat, 256, 1, 0, self.lat0, [0], 512,
                                 flags=SM_CONV_RELU | SM_CONV_RES_ADD, residual=res, res_cstride=512), 2)
        else:
            self.cat = self._buf(B * h0 * w0, 768)
            for l in range(3):
                fh, fw = sizes[l]
                src = self.reg_feat[row0[l]:row0[l] + B * fh * fw]
                self._add("up:cat%d" % l, (lambda s=src, fh=fh, fw=fw, l=l: H.upsample_bilinear(
                    s, self.cat, B, fh, fw, 256, 2 ** l, 256, 768, 256 * l, self.precision == "f32")), 2)
            self._add_conv(_Conv(self, "head.sip_mask_lat0", w_l0, sd[h + "sip_mask_lat0.bias"],
                                 B, [(h0, w0)], [0], self.cat, 768, 1, 0, self.lat0, [0], 512, flags=SM_CONV_RELU), 2)
        self.basis_lo = self._buf(B * h0 * w0, 32, torch.float32)
        self._add_conv(_Conv(self, "head.sip_mask_lat", sd[h + "sip_mask_lat.weight"], sd[h + "sip_mask_lat.bias"], B,
                             [(h0, w0)], [0], self.lat0, 512, 1, 1, self.basis_lo, [0], 32,
                             flags=SM_CONV_RELU | SM_CONV_OUT_F32), 2)
        self.hm, self.wm = 4 * h0, 4 * w0
        # feat_masks [B,Hm,Wm,32] = bilinear x4 of basis_lo (sipmask_head.py:285).  The plan's own mask assembly
        # interpolates AFTER the coefficient dot product (sm_mask_assemble_lo), so the 137 MB tensor is only
        # materialised for the API view (head_outputs) and for the plans that still consume it (_needs_basis)
        self._basis = None
        self._basis_h0w0 = (h0, w0)
        self._add("up:basis", lambda: self._basis_step(), 2)
        # fcos_reg (4, x Scale) + fcos_centerness (1) share reg_feat -> one 5-channel f32 conv
        # (+ 3 zero channels: 8 couts = one 16-byte store per position and the patch kernel's 32-cout tile)
        w_rc = torch.cat([sd[h + "fcos_reg.weight"], sd[h + "fcos_centerness.weight"],
                          torch.zeros_like(sd[h + "fcos_reg.weight"][:3])], 0)
        b_rc = torch.cat([sd[h + "fcos_reg.bias"], sd[h + "fcos_centerness.bias"], torch.zeros_like(sd[h + "fcos_reg.bias"][:3])], 0)
        scales = [float(sd[h + "scales.%d.scale" % i]) for i in range(len(lv))]
        self.reg_out = self._buf(lv.rows, 8, torch.float32)
        self.reg_out.zero_()
        c = self._add_conv(_Conv(self, "head.reg_ctr", w_rc, b_rc, B, sizes, row0, self.reg_feat, 256, 1, 1, self.reg_out,
                                 row0, 8, flags=SM_CONV_OUT_F32 | (_lib.SM_CONV_RELU_NCH if self.benchmark else 0),
                                 scale_nch=4, level_scale=scales))
        c.flops, c.mfma_flops = c.flops * 5 / 8, c.mfma_flops * 5 / 8      # the 3 zero channels are not work (FLOP accounting)
        # FeatureAlign: offset = conv1x1(bbox_pred), y = relu(GN(deform_conv(cls_feat, offset)))
        self.w_off = sd[h + "feat_align.conv_offset.weight"].float().view(72, 4).to(dev).contiguous()
        self.offsets = self._buf(lv.rows, 72, torch.float32)
        self._add("offset", lambda: H.offset_linear(self.reg_out, 8, self.w_off, lv, self.offsets))
        self.aligned = self._buf(lv.rows, 256)
        self._join(1)
        c = self._add_conv(_Conv(self, "head.feat_align", sd[h + "feat_align.conv_adaption.weight"],
                                 sd.get(h + "feat_align.conv_adaption.bias"), B, sizes,
                                 row0, self.cls_feat, 256, 1, 1, self.aligned, row0, 256, deform_groups=4,
                                 offset=self.offsets, flags=(0 if self.flag_norm else SM_CONV_RELU) | _DEFORM_FLAGS))
        # two kernels can run this conv (LDS window / global gather): which one is faster depends on how far the
        # checkpoint's learned offsets reach -- measured once on the first eager run (_tune_deform)
        self._fa_conv = c
        self._deform_tune = (_DEFORM_MODE == "auto" and self.precision == "bf16" and
                             H.deform_conv_window_plan(c.desc) is not None)
        self.deform_choice = None
        if self.flag_norm:                                 # FeatureAlign.forward, sipmask_head.py:49-55
            self._gn("feat_align", self.aligned, sd[h + "feat_align.norm.weight"], sd[h + "feat_align.norm.bias"],
                     conv=c)
        # fcos_cls (80) + sip_cof (128) share the aligned feature -> one 208-channel f32 conv
        w_cc = torch.cat([sd[h + "fcos_cls.weight"], sd[h + "sip_cof.weight"]], 0)
        b_cc = torch.cat([sd[h + "fcos_cls.bias"], sd[h + "sip_cof.bias"]], 0)
        self.ncc = self.ncls + 128
        self.cls_cof = self._buf(lv.rows, self.ncc, torch.float32)
        self._add_conv(_Conv(self, "head.cls_cof", w_cc, b_cc, B, sizes, row0, self.aligned, 256, 1, 1, self.cls_cof,
                             row0, self.ncc, flags=SM_CONV_OUT_F32))
        # VIS track branch (V/...:265-284,310-311): track_convs on levels 0-2 (one 3-level launch per conv) ->
        # bilinear x1/x2/x4 -> cat 768 -> 1x1 -> 512-channel embedding map at stride 8, f32
        self.track_feats = None
        ntrack = sum(1 for k in sd if k.startswith(h + "track_convs.") and k.endswith(".conv.weight"))
        if ntrack:
            lv3 = H.Levels(B, sizes[:3])
            assert list(lv3.row0) == list(row0[:3])
            x = self.pyr
            for i in range(ntrack):
                y = self._buf(lv3.rows, 256)
                name = "track_convs.%d" % i
                c = self._add_conv(_Conv(self, "head." + name, sd[h + name + ".conv.weight"],
                                         None if self.flag_norm else sd.get(h + name + ".conv.bias"), B, sizes[:3],
                                         row0[:3], x, 256, 1, 1, y, row0[:3], 256,
                                         flags=0 if self.flag_norm else SM_CONV_RELU))
                if self.flag_norm and self.precision == "f32":
                    g = sd[h + name + ".gn.weight"].float().to(dev).contiguous()
                    bta = sd[h + name + ".gn.bias"].float().to(dev).contiguous()
                    self._add("gn:" + name, (lambda y=y, g=g, bta=bta: H.groupnorm_f32(
                        y, y, g, bta, self.gn_stats64, lv3, 256, 32, 1e-5, True)))
                elif self.flag_norm:
                    g = sd[h + name + ".gn.weight"].float().to(dev).contiguous()
                    bta = sd[h + name + ".gn.bias"].float().to(dev).contiguous()
                    c.gn_stats = self.gn_stats
                    self._add("gn:" + name, (lambda y=y, g=g, bta=bta: H.groupnorm_apply(
                        y, y, g, bta, self.gn_stats, lv3, 256, 32, 1e-5, True)))
                x = y
            self.track_cat = self._buf(B * h0 * w0, 768)
            for l in range(3):
                fh, fw = sizes[l]
                src = x[row0[l]:row0[l] + B * fh * fw]
                self._add("up:track%d" % l, (lambda s=src, fh=fh, fw=fw, l=l: H.upsample_bilinear(
                    s, self.track_cat, B, fh, fw, 256, 2 ** l, 256, 768, 256 * l, self.precision == "f32")))
            self.track_feats = self._buf(B * h0 * w0, 512, torch.float32)
            self._add_conv(_Conv(self, "head.sipmask_track", sd[h + "sipmask_track.weight"], sd[h + "sipmask_track.bias"],
                                 B, [(h0, w0)], [0], self.track_cat, 768, 1, 0, self.track_feats, [0], 512,
                                 flags=SM_CONV_OUT_F32))

    def _build_post_benchmark(self):
        """SipMaskPostProcessor (B/fcos_core/modeling/rpn/sipmask/inference.py:66-236) for a batch whose images share
        one size: pair selection -> same-label NMS -> top post_top_n -> fused mask assembly."""
        B, lv, bm = self.batch, self.lv, self.benchmark
        npre = int(bm["pre_nms_top_n"])
        kmax = sum(min(npre, h * w * self.ncls) for h, w in lv.sizes)
        self.det_desc = H.make_det_desc(B, lv.sizes, self.strides, lv.row0, self.ncls, self.ncc, 0, self.ncc,
                                        self.ncls, 8, npre, self.img_shape[0], self.img_shape[1], kmax=kmax)
        self.sel = H.pairs_select_alloc(self.det_desc, self.device)
        self.max_num = int(bm["post_top_n"])
        self.nms_out = H.multiclass_nms_alloc(B, kmax, self.ncls, self.max_num, self.device)
        self.box_mul, self.up, (self.ho, self.wo) = H.post_geometry(self.hm, self.wm, self.scale_factor, None)
        sf = float(self.scale_factor)
        self.up = (2.0 / sf, 2.0 / sf)                # crop boxes / 2, upsample 2 / scale_factor (inference.py:205-208)
        import math
        self.ho, self.wo = int(math.floor(self.hm * self.up[0])), int(math.floor(self.wm * self.up[1]))
        self.pitch = (self.wo + 3) // 4 * 4
        self.masks = torch.zeros(B, self.max_num, self.ho, self.pitch, dtype=torch.uint8, device=self.device)
        self.rescorer, self.track_feats = None, None
        self._needs_basis = True
        self._basis = self._buf(B * self.hm * self.wm, 32, torch.float32)
        self._add("det_select", lambda: H.pairs_select(self.det_desc, bm["pre_nms_thresh"], self.cls_cof, self.reg_out,
                                                       self.cls_cof, self.sel))
        self._add("nms", lambda: H.multiclass_nms(self.sel["boxes"], self.sel["scores"], self.sel["ctr"],
                                                  self.sel["ncand"], 0.0, bm["nms_thresh"], self.max_num, self.nms_out))
        self._join(2)
        self._add("mask_assemble", lambda: H.mask_assemble(
            self.basis, True, self.sel["cofs"], self.nms_out["keep"], self.nms_out["det"], self.nms_out["ndet"],
            self.hm, self.wm, self.ho, self.wo, 1.0, 2.0, self.up, 0.4, self.masks))

    def _build_post(self):
        """get_bboxes (sipmask_head.py:500-633) for all images of the batch, device resident."""
        if self.benchmark:
            return self._build_post_benchmark()
        B, lv, cfg = self.batch, self.lv, self.cfg
        self.det_desc = H.make_det_desc(B, lv.sizes, self.strides, lv.row0, self.ncls, self.ncc, 0, self.ncc,
                                        self.ncls, 8, cfg["nms_pre"], self.img_shape[0], self.img_shape[1],
                                        self.scale_factor, bool(self.rescale))
        self.sel = H.det_select_alloc(self.det_desc, self.device)
        # the M/ fast_nms keeps a hard-coded 100 (sipmask_head.py:903), the VIS one cfg.max_per_img (V/...:985)
        self.max_num = 100 if (self.ssd_flag and not self.vis) else cfg["max_per_img"]
        self.nms_out = H.multiclass_nms_alloc(B, self.det_desc.kmax, self.ncls, self.max_num, self.device)
        geo_rescale = (True if self.rescale else None) if self.vis else self.rescale
        self._geo_rescale = geo_rescale
        self.box_mul, up_canvas, (self.ho, self.wo) = H.post_geometry(self.hm, self.wm, self.scale_factor, geo_rescale,
                                                                      self.ssd_flag)
        # scalar up_scale of the launches = the smallest any image may bring: it sizes the kernels' source windows
        _, self.up, _ = H.post_geometry(self.hm, self.wm, self.scale_factor_max, geo_rescale, self.ssd_flag)
        self.up = (min(self.up[0], up_canvas[0]), min(self.up[1], up_canvas[1]))
        self.pitch = (self.wo + 3) // 4 * 4
        # per-image tables (img_shape, scale_factor | crop / upsample geometry), read by det_select / mask assembly /
        # mask_rects; filled with the plan's defaults, rewritten by set_image_metas()
        self.det_tab = torch.zeros(B, 6, dtype=torch.float32, device=self.device)
        self.geom_tab = torch.zeros(B, 8, dtype=torch.float32, device=self.device)
        # ... and result packing: (mask_h, mask_w, canvas_h, canvas_w) of every image (sm_rle_encode_images)
        self.rle_tab = torch.zeros(B, 4, dtype=torch.int32, device=self.device)
        self.det_desc.per_image = self.det_tab.data_ptr()
        self.set_image_metas([dict(img_shape=self.img_shape, scale_factor=self.scale_factor)] * B)
        self.rescorer = None
        if ("bbox_head.convs_scoring.0.conv.weight") in self._sd_keys:
            self.rescorer = MaskRescorer(self._sd, "bbox_head.", B, self.max_num, self.hm, self.wm, self.device)
        # the rescoring branch consumes the cropped probability maps at mask resolution (pos_masks), which only
        # sm_mask_assemble produces; every other plan assembles masks from the conv-resolution basis
        # ... as do geometries whose per-tile source window exceeds the fused kernel's LDS tiles (up_scale = 2 /
        # scale_factor below ~0.45: sm_mask_assemble_lo_supported)
        self.fused_masks = (self.rescorer is None and _FUSED_MASKS
                            and H.mask_assemble_lo_supported(B, self.max_num, 4, self.up))
        self._needs_basis = not self.fused_masks
        if self._needs_basis:
            self._basis = self._buf(B * self.hm * self.wm, 32, torch.float32)     # allocated at build (capture-safe)
        if self.fused_masks:
            self.mask_buf = H.mask_assemble_lo_alloc(B, self.max_num, self.ho, self.wo, self.device)
            self.masks = self.mask_buf["masks"]
        else:
            self.masks = torch.zeros(B, self.max_num, self.ho, self.pitch, dtype=torch.uint8, device=self.device)
        self._add("det_select", lambda: H.det_select(self.det_desc, self.cls_cof, self.reg_out, self.cls_cof, self.sel))
        if self.ssd_flag or self.vis:
            self._add("nms", lambda: H.fast_nms(self.sel["boxes"], self.sel["scores"], self.sel["ctr"],
                                                self.sel["ncand"], cfg["score_thr"], cfg["nms"]["iou_thr"], 200,
                                                self.max_num, self.nms_out))
        else:
            self._add("nms", lambda: H.multiclass_nms(self.sel["boxes"], self.sel["scores"], self.sel["ctr"],
                                                      self.sel["ncand"], cfg["score_thr"], cfg["nms"]["iou_thr"],
                                                      self.max_num, self.nms_out))
        self._join(2)
        pos = None if self.rescorer is None else self.rescorer.pos_masks
        if self.fused_masks:
            h0, w0 = self._basis_h0w0
            self._add("mask_assemble", lambda: H.mask_assemble_lo(
                self.basis_lo, h0, w0, 4, self.sel["cofs"], self.nms_out["keep"], self.nms_out["det"],
                self.nms_out["ndet"], self.ho, self.wo, self.box_mul, 2.0, self.up, self.mask_thr, self.mask_buf,
                per_image=self.geom_tab))
        else:
            self._add("mask_assemble", lambda: H.mask_assemble(
                self._basis, True, self.sel["cofs"], self.nms_out["keep"], self.nms_out["det"], self.nms_out["ndet"],
                self.hm, self.wm, self.ho, self.wo, self.box_mul, 2.0, self.up, self.mask_thr, self.masks, pos,
                per_image=self.geom_tab))
        if self.rescorer is not None:
            self._add("rescore", lambda: self.rescorer.run(self.nms_out["labels"], self.nms_out["det"],
                                                           self.nms_out["ndet"]))
        if self.track_feats is not None:
            # det_roi_feats (V/...:612-616): embeddings at the box centres, boxes back in network coordinates
            import numpy as np
            sfv = float(np.asarray(self.scale_factor, np.float64).reshape(-1)[0])
            self.det_feats = torch.zeros(B, self.max_num, 512, dtype=torch.float32, device=self.device)
            self._add("track_gather", lambda: H.track_gather(
                self.track_feats, self.nms_out["det"], self.nms_out["ndet"], lv.sizes[0][0], lv.sizes[0][1],
                sfv if self.rescale else 1.0, self.det_feats))

    def set_image_metas(self, img_metas, staging=None):
        """img_metas[i]['img_shape'] / ['scale_factor'] of the images of the NEXT run() (the reference reads them per image:
        sipmask_head.py:517-541,579,587-588,621-633).  Two small host->device copies into the tables the kernels read;
        call it outside a captured graph.  Every image's mask (floor(Hm * 2 / scale_factor)) must fit the plan's canvas
        and its scale_factor must not exceed scale_factor_max (prepare() / SipMask.get_masks choose both from the batch).
        staging: (pinned det table, pinned geom table, pinned rle table) -- the copies are then asynchronous on the current stream (the
        caller keeps the pinned pair untouched until they have run: PipelinedPlan.submit)."""
        if self.benchmark:
            raise NotImplementedError("the maskrcnn-benchmark post-processor takes one geometry per plan")
        if len(img_metas) != self.batch:
            raise ValueError("%d img_metas for a plan of %d images" % (len(img_metas), self.batch))
        det, geom, canvas, up_min = H.image_geometry_tables(img_metas, self.hm, self.wm, self._geo_rescale, self.ssd_flag)
        if canvas[0] > self.ho or canvas[1] > self.wo:
            raise ValueError("a mask of %dx%d does not fit the plan's %dx%d canvas: prepare() with the batch's smallest "
                             "scale_factor" % (canvas[0], canvas[1], self.ho, self.wo))
        if up_min[0] < self.up[0] * (1 - 1e-6) or up_min[1] < self.up[1] * (1 - 1e-6):
            raise ValueError("scale_factor above the plan's scale_factor_max (%r): prepare() with the batch's largest" %
                             (self.scale_factor_max,))
        # RLE canvases of the batch: ori_shape with rescale, else img_shape (sipmask_head.py:645-653)
        self.out_hw = [(int(g[4]), int(g[5])) for g in geom]
        self.rle_canvases = [tuple(int(v) for v in (m.get('ori_shape', m['img_shape']) if self.rescale else m['img_shape'])[:2])
                             for m in img_metas]
        rt = torch.tensor([[hw[0], hw[1], cv[0], cv[1]] for hw, cv in zip(self.out_hw, self.rle_canvases)], dtype=torch.int32)
        if staging is not None:
            staging[0].copy_(det)
            staging[1].copy_(geom)
            staging[2].copy_(rt)
            self.det_tab.copy_(staging[0], non_blocking=True)
            self.geom_tab.copy_(staging[1], non_blocking=True)
            self.rle_tab.copy_(staging[2], non_blocking=True)
        else:
            self.det_tab.copy_(det)
            self.geom_tab.copy_(geom)
            self.rle_tab.copy_(rt)
        return self

    def _tune_deform(self):
        """One-off choice of FeatureAlign's kernel from THIS plan's offsets (just produced by the first eager run): the
        LDS-window kernel serves samples within 3 pixels of their tap from LDS and sends a wave with a farther sample
        through a slower global gather; the gather loader costs the same everywhere.  The rule is a function of the DATA --
        the share of offset components beyond the window radius -- not of a timing, so that two plans fed the same input
        choose the same kernel (a timing-based pick differed between otherwise identical plans at small shapes, which
        breaks plan-to-plan bit equality: the two kernels agree only up to accumulation order).  Calibration
        (profiles/r02_deform_conv_microbench.txt, B=2 / B=4 head): offsets ~N(0, 2 px) = 13 % beyond 3 px: window 0.132 /
        0.254 ms vs gather 0.18 / 0.257; ~N(0, 4 px) = 45 %: 0.182 / 0.353 vs 0.18 / 0.258 -> gather above 20 %.
        Both kernels are also timed once on these offsets; the times are reported (bench.py: config.deform_kernel), not used."""
        self._deform_tune = False
        c = self._fa_conv
        far = float((self.offsets.abs() > 3.0).float().mean())          # one device->host read, at the first run only
        if isinstance(c, _DeformChoice):                                # x3 plan: two conv objects instead of a flag
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t = {}
            for name in ("window", "gather"):
                c.pick(name)
                c()
                e0.record()
                for _ in range(3):
                    c()
                e1.record()
                torch.cuda.synchronize()
                t[name] = e0.elapsed_time(e1) / 3
            pick = "gather" if far > 0.20 else "window"
            c.pick(pick)
            self.deform_choice = dict(kernel=pick, offsets_beyond_3px=round(far, 4), window_ms=round(t["window"], 4),
                                      gather_ms=round(t["gather"], 4))
            return
        base = c.desc.flags & ~_lib.SM_CONV_DBG_DEFORM_GATHER
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = {}
        for name, fl in (("window", base), ("gather", base | _lib.SM_CONV_DBG_DEFORM_GATHER)):
            c.desc.flags = fl
            c()
            e0.record()
            for _ in range(3):
                c()
            e1.record()
            torch.cuda.synchronize()
            t[name] = e0.elapsed_time(e1) / 3
        pick = "gather" if far > 0.20 else "window"
        c.desc.flags = base | (_lib.SM_CONV_DBG_DEFORM_GATHER if pick == "gather" else 0)
        c()                                                   # leave the buffers as the chosen kernel writes them
        self.deform_choice = dict(kernel=pick, offsets_beyond_3px=round(far, 4), window_ms=round(t["window"], 4),
                                  gather_ms=round(t["gather"], 4))

    def adopt_deform_choice(self, other):
        """take FeatureAlign's kernel choice of another engine of the same configuration (the slots of a PipelinedPlan, the
        chains of a SubBatchPlan): lazily prepared members see different first batches, and two kernels that agree only up
        to accumulation order would break the slot-to-slot / cut-independent bit equality"""
        if not getattr(self, "_deform_tune", False) or getattr(other, "deform_choice", None) is None:
            return
        self._deform_tune = False
        c = self._fa_conv
        if isinstance(c, _DeformChoice):
            c.pick(other.deform_choice["kernel"])
        else:
            base = c.desc.flags & ~_lib.SM_CONV_DBG_DEFORM_GATHER
            c.desc.flags = base | (_lib.SM_CONV_DBG_DEFORM_GATHER if other.deform_choice["kernel"] == "gather" else 0)
        self.deform_choice = dict(other.deform_choice, adopted=True)

    # -------------------------------------------------------------------------------- execution
    def run(self, img):
        """img: float32 NCHW [B,3,H,W] on the device.  Returns the result dict (device tensors)."""
        assert img.shape == (self.batch, 3, self.H, self.W) and img.dtype == torch.float32 and img.is_cuda
        self.img = img.contiguous()
        if getattr(self, "_deform_tune", False) and not torch.cuda.is_current_stream_capturing():
            self._run_steps(self.steps, self.lanes)           # first eager run: produces the offsets to measure on
            self._tune_deform()
        self._run_steps(self.steps, self.lanes)
        return self.results()

    def results(self):
        o = self.nms_out
        r = dict(det_bboxes=o["det"], det_labels=o["labels"], idxs_keep=o["keep"], ndet=o["ndet"],
                 masks=self.masks[..., :self.wo])
        if self.track_feats is not None:
            r["det_feats"] = self.det_feats
        if self.rescorer is not None:
            r["mask_scores"] = self.rescorer.scores
        return r

    def encode_rle(self, canvas_hw=None, fetch=True, max_runs=8192):
        """Result packing on device (sipmask_head.py:645-657 without the per-mask D2H): run-length encodes the
        masks of the last run() on the current stream, restricted to each detection's box (sm_mask_rects).
        canvas_hw: None = every image's own canvas from its img_metas (img_shape, or ori_shape with rescale); one (H, W) for
        the batch; or a list with one per image.  Images with their own mask size / canvas (a keep_ratio batch) go through
        ONE launch with a per-image table (sm_rle_encode_images).
        Returns per image the list of RLE dicts (fetch=True: two small D2H copies) or the device buffers (then
        `["canvases"]` holds the per-image sizes the dicts need)."""
        per_img = canvas_hw is not None and isinstance(canvas_hw[0], (tuple, list))
        if canvas_hw is None:
            canvases = list(getattr(self, "rle_canvases", None) or [tuple(self.img_shape[:2])] * self.batch)
        elif per_img:
            canvases = [tuple(int(v) for v in c[:2]) for c in canvas_hw]
        else:
            canvases = [tuple(int(v) for v in canvas_hw[:2])] * self.batch
        out_hw = list(getattr(self, "out_hw", None) or [(self.ho, self.wo)] * self.batch)
        uniform = all(c == canvases[0] for c in canvases) and all(hw == (self.ho, self.wo) for hw in out_hw)
        cmax = (max(c[0] for c in canvases), max(c[1] for c in canvases))
        if getattr(self, "_rle", None) is None or self._rle["canvas_w"] < cmax[1] or self._rle["max_runs"] < max_runs:
            self._rle = H.rle_alloc(self.batch, self.max_num, cmax[1], self.device, max_runs=max_runs)
        H.mask_rects(self.nms_out["det"], self.box_mul, 2.0, self.up, self._rle["rect"],
                     per_image=None if self.benchmark else self.geom_tab)
        tab = None
        if not uniform:
            if canvas_hw is None and not self.benchmark:
                tab = self.rle_tab                  # written by set_image_metas (stream-ordered, non-blocking in a pipeline)
            else:
                tab = torch.tensor([[hw[0], hw[1], c[0], c[1]] for hw, c in zip(out_hw, canvases)], dtype=torch.int32).to(self.device)
        H.rle_encode(self.masks, self.nms_out["ndet"], cmax, self._rle, self._rle["rect"], per_image=tab)
        self._rle["canvases"] = canvases
        if not fetch:
            return self._rle
        nd = self.nms_out["ndet"].cpu().tolist()
        try:
            res = H.rle_fetch(self._rle, self.batch, self.max_num, nd, canvases)
        except RuntimeError:
            need = int(-self._rle["nruns"].min().item()) + 1
            if need <= max_runs:
                raise
            return self.encode_rle(canvas_hw, True, need)
        return res

    # -------------------------------------------------------------------------------- API views
    def head_outputs(self):
        """(cls_scores, bbox_preds, centernesses, cof_preds, feat_masks) as NCHW views, as
        SipMaskHead.forward returns them (sipmask_head.py:287)."""
        B, lv = self.batch, self.lv
        cls, bb, ctr, cof = [], [], [], []
        for l, (h, w) in enumerate(lv.sizes):
            r0, n = lv.row0[l], B * h * w
            cc = self.cls_cof[r0:r0 + n].view(B, h, w, self.ncc).permute(0, 3, 1, 2)
            rr = self.reg_out[r0:r0 + n].view(B, h, w, 8).permute(0, 3, 1, 2)
            cls.append(cc[:, :self.ncls])
            cof.append(cc[:, self.ncls:])
            bb.append(rr[:, :4] * float(self.strides[l]))
            ctr.append(rr[:, 4:5])
        fm = self.basis.view(B, self.hm, self.wm, 32).permute(0, 3, 1, 2)
        return cls, bb, ctr, cof, fm

    def total_conv_flops(self):
        return sum(c.flops for c in self.convs) + sum(c.flops for c in self.fused) + getattr(self, "stem_flops", 0.0)


class SubBatchPlan:
    """A batch run as `lanes` independent sub-batches, each with its own launch plan, on concurrent HIP streams (one
    linear launch chain per sub-batch, forked from and joined to the caller's stream; capture-safe).

    Why: a launch plan is ~125 dependent kernels, and at 4 images per step a third of the step is what happens BETWEEN
    them (launch boundaries, ramp-up, tails, single-tile K loops of the small layers: t(B) = 1.65 ms + 0.87 ms/image,
    profiles/r02*).  Two chains of B/2 fill each other's gaps: measured 887 vs 850 img/s at B=4, 1046 vs 912 at B=8
    (MI355X, hipGraph replay).  Sub-batches of ONE image lose more per launch than they gain (648 img/s), hence the
    `lanes="auto"` rule of SipMask.prepare: 2 lanes iff the batch is even and >= 4.
    The sub-plans run without their own side lanes: forking side streams from a stream that is itself a fork
    crashes hipStreamEndCapture on ROCm 7.2 (segfault), and the second chain covers what the side lanes covered.
    Outputs are ONE set of batch-sized tensors; every sub-plan writes its slice."""

    def __init__(self, engines):
        assert len(engines) >= 2 and all(e.rescorer is None for e in engines)
        self.engines = engines
        self.batch = sum(e.batch for e in engines)
        e0 = engines[0]
        self.lv, self.ncls, self.cfg, self.max_num = e0.lv, e0.ncls, e0.cfg, e0.max_num
        self.H, self.W, self.ho, self.wo, self.pitch = e0.H, e0.W, e0.ho, e0.wo, e0.pitch
        dev = e0.device
        mk = lambda t: torch.zeros((self.batch,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        self.out = {k: mk(e0.nms_out[k]) for k in ("det", "labels", "keep", "ndet")}
        self.masks = mk(e0.masks)
        self.det_feats = mk(e0.det_feats) if e0.track_feats is not None else None
        b0 = 0
        for e in engines:                       # every sub-plan writes its slice of the shared outputs
            e.multi_stream = False
            sl = slice(b0, b0 + e.batch)
            for k in self.out:
                e.nms_out[k] = self.out[k][sl]
            e.masks = self.masks[sl]
            if getattr(e, "fused_masks", False):
                e.mask_buf["masks"] = e.masks
            if self.det_feats is not None:
                e.det_feats = self.det_feats[sl]
            b0 += e.batch
        self.streams = [torch.cuda.Stream(device=dev) for _ in engines[1:]]
        self.out_hw = [hw for e in engines for hw in getattr(e, "out_hw", [])]

    def set_image_metas(self, img_metas):
        """per-image img_shape / scale_factor (SipMaskEngine.set_image_metas) for every chain's slice of the batch"""
        b0 = 0
        for e in self.engines:
            e.set_image_metas(img_metas[b0:b0 + e.batch])
            b0 += e.batch
        self.out_hw = [hw for e in self.engines for hw in e.out_hw]
        return self

    def run(self, img):
        assert img.shape[0] == self.batch
        main = torch.cuda.current_stream()
        e0 = self.engines[0]
        if getattr(e0, "_deform_tune", False) and not torch.cuda.is_current_stream_capturing():
            e0.run(img[:e0.batch])                      # chain 0 chooses FeatureAlign's kernel, the other chains adopt it
            for e in self.engines[1:]:
                e.adopt_deform_choice(e0)
        b0 = self.engines[0].batch
        for e, st in zip(self.engines[1:], self.streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                e.run(img[b0:b0 + e.batch])
            b0 += e.batch
        self.engines[0].run(img[:self.engines[0].batch])
        for st in self.streams:
            main.wait_stream(st)
        return self.results()

    def capture(self, img, multi_stream=True):
        """ONE hipGraph PER SUB-PLAN instead of one around run(): a sub-plan captured on its own may keep its internal
        side lanes (no fork of a fork inside one capture), and replay() launches the graphs on concurrent streams.
        `img` must stay where it is (the graphs read its slices)."""
        assert img.shape[0] == self.batch
        self.graphs, b0 = [], 0
        for e in self.engines:
            e.multi_stream = bool(multi_stream)
            sub = img[b0:b0 + e.batch]
            if e is not self.engines[0]:
                e.adopt_deform_choice(self.engines[0])
            e.run(sub)                                   # eager once with this lane setting (side streams get created)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                e.run(sub)
            self.graphs.append(g)
            b0 += e.batch
        return self

    def replay(self, join=True):
        """join=True: fork from / join to the caller's stream around every replay (results valid in stream order).
        join=False: FREE-RUNNING chains -- sub-plan i replays on its own stream behind its previous replay only, so
        the chains of consecutive batches drift apart and one chain's low-parallelism tail (top-k select, NMS on a few
        blocks, mask assembly) overlaps the other chain's convs of the NEXT batch instead of the other chain's tail.
        The caller calls join() before reading the outputs or overwriting the input."""
        main = torch.cuda.current_stream()
        if not join:
            if not hasattr(self, "free_streams"):
                self.free_streams = [torch.cuda.Stream(device=self.engines[0].device) for _ in self.engines]
            for g, st in zip(self.graphs, self.free_streams):
                st.wait_stream(main)                    # the input the caller produced on its stream
                with torch.cuda.stream(st):
                    g.replay()
            return None
        for g, st in zip(self.graphs[1:], self.streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                g.replay()
        self.graphs[0].replay()
        for st in self.streams:
            main.wait_stream(st)
        return self.results()

    def join(self):
        """the caller's stream waits for every free-running chain (replay(join=False))"""
        main = torch.cuda.current_stream()
        for st in getattr(self, "free_streams", ()):
            main.wait_stream(st)
        return self.results()

    def offset_chains(self):
        """One-time phase offset for free-running chains: chain i > 0 first runs i/n of a pass eagerly on its stream (its
        outputs are overwritten by the next replay), so the chains start out of phase instead of in lock step."""
        main = torch.cuda.current_stream()
        if not hasattr(self, "free_streams"):
            self.free_streams = [torch.cuda.Stream(device=self.engines[0].device) for _ in self.engines]
        n = len(self.engines)
        for i, (e, st) in enumerate(zip(self.engines, self.free_streams)):
            if i == 0:
                continue
            st.wait_stream(main)
            with torch.cuda.stream(st):
                k = len(e.steps) * i // n
                e._run_steps(e.steps[:k], e.lanes[:k])

    def results(self):
        r = dict(det_bboxes=self.out["det"], det_labels=self.out["labels"], idxs_keep=self.out["keep"], ndet=self.out["ndet"],
                 masks=self.masks[..., :self.wo])
        if self.det_feats is not None:
            r["det_feats"] = self.det_feats
        return r

    def encode_rle(self, canvas_hw=None, fetch=True, max_runs=8192):
        assert fetch, "device-side RLE buffers are per sub-plan"
        per_img = canvas_hw is not None and isinstance(canvas_hw[0], (tuple, list))
        out, b0 = [], 0
        for e in self.engines:
            out.extend(e.encode_rle(canvas_hw[b0:b0 + e.batch] if per_img else canvas_hw, True, max_runs))
            b0 += e.batch
        return out

    @property
    def convs(self):
        return [c for e in self.engines for c in e.convs]

    def total_conv_flops(self):
        return sum(e.total_conv_flops() for e in self.engines)


class PostProcessor:
    """SipMaskHead.get_bboxes on caller-provided head outputs (the API-faithful path used by the
    parity tests: identical f32 inputs on both sides).  sipmask_head.py:500-633."""

    def __init__(self, cls_scores, bbox_preds, centernesses, cof_preds, feat_masks, img_metas, cfg, strides,
                 rescale=None, ssd_flag=False, vis=False, rescore_sd=None):
        _lib.load()
        dev = cls_scores[0].device
        _lib.require_cuda(cls_scores[0], feat_masks)
        B = cls_scores[0].shape[0]
        assert len(img_metas) == B
        self.B, self.dev = B, dev
        sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        lv = H.Levels(B, sizes)
        C = cls_scores[0].shape[1]
        rows = lambda ts: torch.cat([t.detach().float().permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts])
        self.cls = rows(cls_scores).contiguous()
        self.cof = rows(cof_preds).contiguous()
        self.reg = torch.cat([rows(bbox_preds), rows(centernesses),
                              torch.zeros(lv.rows, 3, device=dev)], 1).contiguous()
        self.basis = feat_masks.detach().float().contiguous()           # [B,32,Hm,Wm]
        self.hm, self.wm = self.basis.shape[-2:]
        # every image brings its own img_shape / scale_factor (sipmask_head.py:517-541: get_bboxes_single is called with
        # img_metas[img_id]): the kernels read them from two small per-image device tables
        self.cfg, self.ssd_flag, self.vis = cfg, bool(ssd_flag), bool(vis)
        self.mask_thr = 0.5 if self.vis else 0.4                      # V/...:764 vs sipmask_head.py:633
        geo_rescale = (True if rescale else None) if self.vis else rescale
        det_tab, geom_tab, (self.ho, self.wo), self.up = H.image_geometry_tables(img_metas, self.hm, self.wm, geo_rescale,
                                                                                self.ssd_flag)
        self.det_tab, self.geom_tab = det_tab.to(dev), geom_tab.to(dev)
        self.out_hw = [(int(g[4]), int(g[5])) for g in geom_tab]       # (Ho, Wo) of every image
        self.box_mul = (float(geom_tab[0, 0]), float(geom_tab[0, 1]))  # scalars: unused while the tables are given
        self.pitch = (self.wo + 3) // 4 * 4
        meta = img_metas[0]
        self.desc = H.make_det_desc(B, sizes, strides, lv.row0, C, C, 0, 128, 0, 8, cfg.get('nms_pre', -1),
                                    meta['img_shape'][0], meta['img_shape'][1], meta.get('scale_factor', 1.0), bool(rescale), True)
        self.desc.per_image = self.det_tab.data_ptr()
        self.sel = H.det_select_alloc(self.desc, dev)
        self.max_num = 100 if (self.ssd_flag and not self.vis) else cfg['max_per_img']
        self.out = H.multiclass_nms_alloc(B, self.desc.kmax, C, self.max_num, dev)
        # SipMask++: rescore_sd = the head's state_dict (keys convs_scoring.*, mask_scoring.*)
        self.rescorer = None if rescore_sd is None else MaskRescorer(rescore_sd, "", B, self.max_num, self.hm, self.wm, dev)

    def run(self, want_pos_masks=False):
        cfg = self.cfg
        H.det_select(self.desc, self.cls, self.reg, self.cof, self.sel)
        if self.ssd_flag or self.vis:
            H.fast_nms(self.sel["boxes"], self.sel["scores"], self.sel["ctr"], self.sel["ncand"], cfg['score_thr'],
                       cfg['nms']['iou_thr'], 200, self.max_num, self.out)
        else:
            H.multiclass_nms(self.sel["boxes"], self.sel["scores"], self.sel["ctr"], self.sel["ncand"],
                             cfg['score_thr'], cfg['nms']['iou_thr'], self.max_num, self.out)
        masks = torch.zeros(self.B, self.max_num, self.ho, self.pitch, dtype=torch.uint8, device=self.dev)
        if self.rescorer is not None:
            self.pos_masks = self.rescorer.pos_masks
            self.pos_masks.zero_()
        else:
            self.pos_masks = (torch.zeros(self.B, self.max_num, self.hm, self.wm, device=self.dev)
                              if want_pos_masks else None)
        H.mask_assemble(self.basis, False, self.sel["cofs"], self.out["keep"], self.out["det"], self.out["ndet"],
                        self.hm, self.wm, self.ho, self.wo, self.box_mul, 2.0, self.up, self.mask_thr, masks,
                        self.pos_masks, per_image=self.geom_tab)
        self.masks = masks
        self.mask_scores = None
        if self.rescorer is not None:
            self.mask_scores = self.rescorer.run(self.out["labels"], self.out["det"], self.out["ndet"])
        nd = self.out["ndet"].cpu().tolist()
        res = []
        for b in range(self.B):
            n = nd[b]
            ho, wo = self.out_hw[b]                    # this image's mask size (planes share the batch's canvas)
            res.append((self.out["det"][b, :n], self.out["labels"][b, :n], self.out["keep"][b, :n],
                        masks[b, :n, :ho, :wo]))
        return res

    def encode_rle(self, canvas_hw):
        """RLE dicts of the masks of the last run(), per image (sipmask_head.py:645-657), encoded on device.
        canvas_hw: one (H, W) for the batch or a list with one per image (img_metas[i]['ori_shape' / 'img_shape'])."""
        from . import ops as P
        rect = torch.zeros(self.B * self.max_num, 4, dtype=torch.int32, device=self.dev)
        H.mask_rects(self.out["det"], self.box_mul, 2.0, self.up, rect, per_image=self.geom_tab)
        per_img = isinstance(canvas_hw[0], (tuple, list))
        same = not per_img and all(hw == (self.ho, self.wo) for hw in self.out_hw)
        if same:                                       # one geometry: the whole batch in one launch
            return P.encode_masks(self.masks[..., :self.wo], self.out["ndet"], canvas_hw, rect)
        out = []
        for b in range(self.B):                        # result packing is outside the timed path: one launch per image
            ho, wo = self.out_hw[b]
            cv = canvas_hw[b] if per_img else canvas_hw
            out += P.encode_masks(self.masks[b:b + 1, :, :ho, :wo].contiguous(), self.out["ndet"][b:b + 1], cv,
                                  rect.view(self.B, self.max_num, 4)[b].contiguous())
        return out


class PipelinedPlan:
    """Several STEPS in flight.  `plans` are complete launch plans of one configuration -- each with its own activation
    buffers, outputs, static input and hipGraph -- used round-robin on their own streams: submit(img) enqueues the next
    step on the next slot and returns at once, so step k+1's backbone runs while step k is still in its low-parallelism
    tail (top-k select, NMS on a few blocks, mask assembly) and through the under-filled launches of layer3 / layer4 /
    the small FPN levels.  Inside one step the same overlap had to come from cutting the batch in two (SubBatchPlan),
    i.e. from kernels half the size: measured on R50 800 x 1344, B=4 per step (tools/pipeline_steps_bench.py): one step
    in flight as two B=2 chains 1 004-1 009 img/s; two steps in flight, one B=4 chain each, 1 217-1 225; three 1 266.
    Results are the slot's own tensors: read them (results(slot)) before `depth` further submits reuse the slot.
    A throughput structure: the latency of one step is that of the single plan (4.1 ms at B=4), not lower."""

    def __init__(self, plans, graph=True):
        assert len(plans) >= 1
        self.plans = list(plans)
        self.depth = len(self.plans)
        self.batch = self.plans[0].batch
        dev = self.plans[0].device if hasattr(self.plans[0], "device") else self.plans[0].engines[0].device
        self.device = dev
        self.use_graph = bool(graph)
        for p in self.plans:                               # the other steps in flight are what fills a launch's idle CUs: no side
            if hasattr(p, "multi_stream"):                 # lanes inside a slot (1 217 vs 1 147 img/s with them)
                p.multi_stream = False
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.plans]
        self.static = [None] * self.depth
        self.graphs = [None] * self.depth
        self.done = [None] * self.depth
        self.next_slot = 0
        self.last_slot = None
        self._host = {}                                    # per slot: two sets of pinned host buffers of submit(pack=True)
        self._packs, self._pack_gen = {}, {}               # per slot: unread result sets (oldest first), packs issued
        self._rle_sets = {}                                # per slot: two device-side RLE buffer sets used alternately
        self._pack_prefix = {}                             # per slot: bytes of RLE strings that travel with the step (adaptive)
        self._meta_pins, self._meta_gen = {}, {}           # per slot: two pinned (det, geom) table pairs + their copy events
        self.last_fetch = dict(wait_s=0.0, pack_s=0.0)     # host seconds of the last fetch(): waiting for the GPU / building dicts

    def _prepare_slot(self, k, img):
        plan = self.plans[k]
        static = img.clone()
        if k > 0 and self.static[0] is None:               # slot 0 chooses FeatureAlign's kernel for every slot
            self._prepare_slot(0, img)
        if k > 0:
            for a, b in zip(getattr(plan, "engines", [plan]), getattr(self.plans[0], "engines", [self.plans[0]])):
                a.adopt_deform_choice(b)
        plan.run(static)                                   # eager once: lazy buffers, the deformable-kernel choice
        torch.cuda.synchronize(self.device)
        if self.use_graph:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                plan.run(static)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                plan.run(static)
            self.graphs[k] = g
        self.static[k] = static

    def capture(self, img):
        """builds every slot's static input (and hipGraph) from an example batch; called by the first submit otherwise"""
        for k in range(self.depth):
            if self.static[k] is None:
                self._prepare_slot(k, img)
        return self

    def submit(self, img, img_metas=None, pack=False, canvas_hw=None, max_runs=8192):
        """enqueue one step on the next slot (after whatever produced `img` on the caller's stream); returns the slot.
        img_metas: this BATCH's per-image img_shape / scale_factor -- written into the slot's own device tables on the
        slot's stream, i.e. behind the slot's previous step and in front of this one (the other slots, whose steps may
        still be in flight, keep the metas they were submitted with; the reference reads img_metas per batch,
        sipmask_head.py:517-541).  pack=True: result packing rides on the slot's stream behind the step (sm_mask_rects +
        sm_rle_encode, then the boxes / labels / counts / RLE strings go to pinned host buffers asynchronously): fetch(slot)
        returns them (sipmask_head.py:645-662, M/mmdet/apis/test.py:12-72 return results per batch)."""
        assert img.shape[0] == self.batch
        k = self.next_slot
        if pack and len(self._packs.get(k, ())) >= 2:       # checked BEFORE anything is enqueued
            raise RuntimeError("PipelinedPlan: slot %d has two unread result sets -- fetch(slot) at least every other submit" % k)
        if self.static[k] is None:
            self._prepare_slot(k, img)
        st = self.streams[k]
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            if img_metas is not None:
                self._submit_metas(k, img_metas)
            self.static[k].copy_(img, non_blocking=True)
            if self.graphs[k] is not None:
                self.graphs[k].replay()
            else:
                self.plans[k].run(self.static[k])
            if pack:
                self._pack(k, canvas_hw, max_runs)
            ev = torch.cuda.Event()
            ev.record(st)
        self.done[k] = ev
        self.last_slot = k
        self.next_slot = (k + 1) % self.depth
        return k

    def _submit_metas(self, k, img_metas):
        """this batch's img_metas into slot k's device tables, on the slot's stream and WITHOUT blocking the host: the tables
        are staged in pinned memory (two pairs per slot, used alternately; a pair is refilled only after the copy that read it
        has run -- by then `depth` further submits have passed) and copied with non_blocking=True.  (A pageable source made
        every submit wait for the slot's previous step: ADVICE r4.)"""
        plan = self.plans[k]
        engs = getattr(plan, "engines", None)
        if engs is not None:                                # SubBatchPlan slots keep the blocking path (not the timed structure)
            plan.set_image_metas(img_metas)
            return
        pins = self._meta_pins.setdefault(k, [])
        g = self._meta_gen.get(k, 0)
        self._meta_gen[k] = g + 1
        if len(pins) < 2:
            pins.append(dict(det=torch.empty(plan.det_tab.shape, dtype=plan.det_tab.dtype, pin_memory=True),
                             geom=torch.empty(plan.geom_tab.shape, dtype=plan.geom_tab.dtype, pin_memory=True),
                             rle=torch.empty(plan.rle_tab.shape, dtype=plan.rle_tab.dtype, pin_memory=True), ev=None))
            pin = pins[-1]
        else:
            pin = pins[g % 2]
            if pin["ev"] is not None:
                pin["ev"].synchronize()                     # two submits of this slot ago: long done
        plan.set_image_metas(img_metas, staging=(pin["det"], pin["geom"], pin["rle"]))
        pin["ev"] = torch.cuda.Event()
        pin["ev"].record(self.streams[k])

    # RLE strings of one batch that travel with the first (asynchronous) copy; a batch with longer strings pays a second,
    # synchronous copy in fetch() (from the slot's device buffers, which are double-buffered like the pinned sets).  The
    # prefix ADAPTS (round 6): it starts at PACK_PREFIX_MIN and grows to twice the longest batch seen (a power of two, at
    # most PACK_PREFIX_BYTES) -- a step's strings are 30-150 KB, and copying a fixed 4 MB per step was 1.4 GB/s of PCIe
    # traffic at BASELINE's shape and 16 GB/s in the six-slot stress configuration.
    PACK_PREFIX_BYTES = 4 << 20
    PACK_PREFIX_MIN = 256 << 10

    def _pack(self, k, canvas_hw, max_runs=8192):
        """on the slot's stream, behind its step: device-side RLE of the step's masks + asynchronous D2H of everything the
        caller's evaluation loop consumes (boxes, labels, counts, run counts, string offsets, a prefix of the strings).
        Every slot owns TWO sets of pinned host buffers used alternately, so the host may submit the slot's next step
        BEFORE it has read this one's results (fetch() then runs while `depth` steps are in flight; with one set the host
        had to wait for the slot, read it, and only then resubmit -- one step fewer in flight during every read: measured
        1 216-1 244 vs 1 330 img/s without results)."""
        plan = self.plans[k]
        if hasattr(plan, "engines"):
            raise NotImplementedError("submit(pack=True): single-chain slots (det.prepare(..., in_flight=N) builds them)")
        q = self._packs.setdefault(k, [])
        # two device-side RLE buffer sets per slot, alternated like the pinned sets: the strings of step k stay on the device
        # until the slot's second next pack, so fetch()'s fallback copy of an over-long batch never reads rewritten bytes
        sets_d = self._rle_sets.setdefault(k, [None, None])
        gd = self._pack_gen.get(k, 0) % 2
        plan._rle = sets_d[gd]
        rle = plan.encode_rle(canvas_hw, fetch=False, max_runs=max_runs)
        sets_d[gd] = plan._rle
        sets = self._host.setdefault(k, [])
        o = plan.nms_out
        if len(sets) < 2:
            pin = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            sets.append(dict(det=pin(o["det"]), labels=pin(o["labels"]), ndet=pin(o["ndet"]), nruns=pin(rle["nruns"]),
                             offsets=pin(rle["offsets"]),
                             packed=torch.empty(min(self.PACK_PREFIX_BYTES, rle["packed"].numel()), dtype=torch.uint8,
                                                pin_memory=True)))
            hb = sets[-1]
        else:
            hb = sets[self._pack_gen.get(k, 0) % 2]
        self._pack_gen[k] = self._pack_gen.get(k, 0) + 1
        prefix = min(hb["packed"].numel(), self._pack_prefix.get(k, self.PACK_PREFIX_MIN))
        # ONE launch writes all six pieces into the pinned set (sm_copy_segments) instead of six hipMemcpyAsync calls: one
        # host call per step on the submit path.  (Round 6 first blamed those copies for the pipeline's rare GPU memory fault --
        # with them the six-slot stress configuration faulted ten times as often -- until the cause turned out to be a barrier
        # race in the NMS sort, whose window their traffic widened: DESIGN section 6.)
        H.copy_segments([(o["det"], hb["det"]), (o["labels"], hb["labels"]), (o["ndet"], hb["ndet"]), (rle["nruns"], hb["nruns"]),
                         (rle["offsets"], hb["offsets"]), (rle["packed"][:prefix], hb["packed"][:prefix])])
        ev = torch.cuda.Event()
        ev.record(self.streams[k])
        q.append(dict(ev=ev, hb=hb, canvases=list(rle["canvases"]), rle=rle, prefix=prefix))

    def fetch(self, slot=None):
        """the packed results of the OLDEST unread step submitted to `slot` with pack=True: blocks the HOST until that step
        (and its copies) are done, returns per image (det_bboxes [n,5] ndarray, det_labels [n] ndarray, [RLE dict] * n).
        A slot holds at most two unread result sets."""
        import time as _time
        k = self.last_slot if slot is None else slot
        q = self._packs.get(k)
        if not q:
            raise RuntimeError("PipelinedPlan.fetch: no unread results on slot %r (submit(..., pack=True) first)" % (k,))
        rec = q[0]                              # popped only once nothing can fail any more
        t0 = _time.perf_counter()
        rec["ev"].synchronize()
        t1 = _time.perf_counter()
        hb = rec["hb"]
        nruns, offs = hb["nruns"].numpy(), hb["offsets"].numpy()
        if (nruns < 0).any():
            q.pop(0)                            # this result set is unusable whatever the caller does next
            raise RuntimeError("sm_rle_encode: max_runs too small, a mask needs %d runs" % (-nruns.min()))
        total = int(offs[-1])
        prefix = rec["prefix"]
        blob = hb["packed"][:min(total, prefix)].numpy().tobytes()
        if total > prefix:                      # rare: longer strings than the prefix that travelled with the step -- the
            # slot's device buffers are double-buffered (_pack), and a slot holds at most two unread sets: still intact
            blob += rec["rle"]["packed"][prefix:total].cpu().numpy().tobytes()
            grow = 1 << int(math.ceil(math.log2(2 * total)))          # the next packs of this slot carry twice this batch
            self._pack_prefix[k] = min(self.PACK_PREFIX_BYTES, max(grow, self._pack_prefix.get(k, self.PACK_PREFIX_MIN)))
        q.pop(0)
        plan = self.plans[k]
        out, mx = [], plan.max_num
        nd = hb["ndet"].numpy()
        for b in range(self.batch):
            n = int(nd[b])
            size = [int(rec["canvases"][b][0]), int(rec["canvases"][b][1])]
            out.append((hb["det"][b, :n].numpy().copy(), hb["labels"][b, :n].numpy().copy(),
                        [dict(size=list(size), counts=blob[offs[b * mx + i]:offs[b * mx + i + 1]]) for i in range(n)]))
        self.last_fetch = dict(wait_s=t1 - t0, pack_s=_time.perf_counter() - t1)
        return out

    def results(self, slot=None):
        """the step's outputs, valid in the caller's stream order (the caller's stream waits for the slot)"""
        k = self.last_slot if slot is None else slot
        if self.done[k] is not None:
            torch.cuda.current_stream().wait_event(self.done[k])
        return self.plans[k].results()

    @property
    def out_hw(self):
        """(Ho, Wo) of every image of the LAST submitted step (each slot keeps the metas it was submitted with)"""
        return self.plans[self.last_slot if self.last_slot is not None else 0].out_hw

    def join(self):
        for ev in self.done:
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
        return self

    def run(self, img, img_metas=None):
        """one step, start to finish (the plain plan interface)"""
        return self.results(self.submit(img, img_metas))

    def set_image_metas(self, img_metas):
        """default metas of EVERY slot (e.g. right after prepare()).  Steps still in flight read their slot's tables, so all
        slots are joined first; per-batch metas belong in submit(img, img_metas) (the tables of a slot must
        never change under a step in flight)."""
        for ev in self.done:
            if ev is not None:
                ev.synchronize()
        for p in self.plans:
            p.set_image_metas(img_metas)
        return self

    def encode_rle(self, *a, slot=None, **kw):
        k = self.last_slot if slot is None else slot
        self.results(k)
        return self.plans[k].encode_rle(*a, **kw)

    @property
    def convs(self):
        return self.plans[0].convs

    def total_conv_flops(self):
        return self.plans[0].total_conv_flops()

