"""Runs the reference's own pure-Python / PyTorch code in THIS container (build box only; /root/reference does not
travel).  Used by tests/golden/make_reference_vectors.py to produce the committed fixtures; never imported by tests.

The reference is mmdetection v1.1 (M/ = SipMask-mmdetection, V/ = SipMask-VIS) and maskrcnn-benchmark (B/): its
packages need mmcv, pycocotools and compiled CUDA/C++ extensions, none of which exist here, so `import mmdet` fails.
What this loader does instead: it registers *stub* modules for exactly those absent third-party / compiled pieces
and then executes the reference's source files, unmodified and from where they lie, under their own module names.
Everything that is plain PyTorch in the reference (target assignment, point grids, box decoding, the loss
assembly, fast_nms, the python crop_split, the tracking scores, the post-processing control flow) therefore runs as
written.  What is replaced, and by what, is listed in STAND_INS and written into the fixture file.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

# /root/reference is read-only by contract.  Executing its source files through importlib would otherwise drop
# __pycache__/*.pyc next to them (root bypasses the mount's permission bits); this switch must precede every load().
sys.dont_write_bytecode = True

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ops as O  # noqa: E402

REF = "/root/reference"
M = os.path.join(REF, "SipMask-mmdetection")
V = os.path.join(REF, "SipMask-VIS")
B = os.path.join(REF, "SipMask-benchmark")

STAND_INS = {
    "mmcv": "absent third-party package: normal_init / kaiming_init / constant_init / xavier_init re-stated from their "
            "published definitions (torch.nn.init calls); nothing else of mmcv is reached by the captured functions",
    "pycocotools.mask.encode": "absent third-party package: replaced by the identity (the binary mask is returned, so "
                               "the fixture holds the reference's mask, not its RLE)",
    "mmdet.ops.nms.nms_wrapper.nms": "compiled extension: replaced by oracle.ops.nms(mode='gpu'), which reproduces the "
                                     "reference's own NMS golden vectors (tests/golden/nms_kat.json)",
    "mmdet.ops.DeformConv": "compiled CUDA extension: replaced by oracle.ops.deform_conv (parity unpinned for that op); "
                            "DeformConvPack = the same behind its conv_offset conv, as deform_conv.py:258-296 defines it",
    "mmdet.ops.CropSplit / CropSplitGt": "compiled CUDA extension: replaced by oracle.ops.crop_split / crop_split_gt "
                                          "(cross-checked in the fixtures against the reference's python crop_split)",
    "mmdet.ops.sigmoid_focal_loss": "compiled CUDA extension: replaced by the reference's own py_sigmoid_focal_loss "
                                    "(mmdet/models/losses/focal_loss.py), the formula the CUDA kernel implements",
}


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []            # behaves as a package for "from x.y import z"
    sys.modules[name] = m
    return m


def load(name, path):
    """execute the reference source file `path` as module `name` (parents must already be registered)"""
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    parent, _, leaf = name.rpartition(".")
    if parent in sys.modules:
        setattr(sys.modules[parent], leaf, m)
    return m


class _Registry:
    def __init__(self):
        self.module_dict = {}

    def register_module(self, cls):
        self.module_dict[cls.__name__] = cls
        return cls


def _force_fp32(apply_to=None, out_fp16=False):
    return lambda f: f


def _nms_stub(dets, iou_thr, device_id=None):
    """signature of mmdet/ops/nms/nms_wrapper.py:nms -> (dets[inds], inds)"""
    d = dets.detach().cpu().numpy().astype(np.float32)
    keep = O.nms(d, iou_thr, mode="gpu")
    inds = torch.as_tensor(np.asarray(keep, dtype=np.int64))
    return dets[inds], inds


class _DeformConv(nn.Module):
    """constructor signature of mmdet/ops/dcn/deform_conv.py:DeformConv"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert groups == 1 and not bias
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, k, k))
        self.stride, self.padding, self.dilation, self.dg = stride, padding, dilation, deformable_groups

    def forward(self, x, offset):
        return O.deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.dg)


class _DeformConvPack(_DeformConv):
    """M/mmdet/ops/dcn/deform_conv.py:258-296 restated around the stand-in: conv_offset (a plain conv with the layer's own
    kernel / stride / padding, bias) feeds the deformable conv"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, bias)
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.conv_offset = nn.Conv2d(in_channels, deformable_groups * 2 * k * k, kernel_size=k, stride=stride,
                                     padding=padding, bias=True)

    def forward(self, x):
        return super().forward(x, self.conv_offset(x))


class _CropSplit(nn.Module):
    def __init__(self, c=2):
        super().__init__()
        self.c = c

    def forward(self, data, rois):
        out = O.crop_split(data.detach().numpy(), rois.detach().numpy().astype(np.float32), self.c)
        return torch.from_numpy(out)


class _CropSplitGt(nn.Module):
    def __init__(self, c=2):
        super().__init__()

    def forward(self, data, rois):
        return torch.from_numpy(O.crop_split_gt(data.detach().numpy(), rois.detach().numpy().astype(np.float32)))


def _mmcv():
    def normal_init(module, mean=0, std=1, bias=0):
        nn.init.normal_(module.weight, mean, std)
        if getattr(module, "bias", None) is not None:
            nn.init.constant_(module.bias, bias)

    def constant_init(module, val, bias=0):
        nn.init.constant_(module.weight, val)
        if getattr(module, "bias", None) is not None:
            nn.init.constant_(module.bias, bias)

    def kaiming_init(module, a=0, mode="fan_out", nonlinearity="relu", bias=0, distribution="normal"):
        (nn.init.kaiming_uniform_ if distribution == "uniform" else nn.init.kaiming_normal_)(
            module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
        if getattr(module, "bias", None) is not None:
            nn.init.constant_(module.bias, bias)

    def xavier_init(module, gain=1, bias=0, distribution="normal"):
        (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(module.weight, gain=gain)
        if getattr(module, "bias", None) is not None:
            nn.init.constant_(module.bias, bias)

    cnn = stub("mmcv.cnn", normal_init=normal_init, constant_init=constant_init, kaiming_init=kaiming_init,
               xavier_init=xavier_init)
    stub("mmcv", cnn=cnn)


_loaded = {}


def mmdet_tree(root=M, tag="M"):
    """Registers the stubs and executes the pure-Python part of the mmdet tree under `root`.  Returns a namespace of
    the loaded reference modules.  M/ and V/ are two different mmdet trees: call with a fresh interpreter per tree, or
    rely on the returned namespace (sys.modules entries of the previous tree are dropped)."""
    for k in [k for k in sys.modules if k == "mmdet" or k.startswith("mmdet.") or k.startswith("mmcv") or
              k.startswith("pycocotools")]:
        del sys.modules[k]
    _mmcv()
    stub("pycocotools", mask=stub("pycocotools.mask", encode=lambda a: [a]))
    p = os.path.join(root, "mmdet")
    stub("mmdet")
    stub("mmdet.core")
    stub("mmdet.core.bbox")
    stub("mmdet.core.utils")
    stub("mmdet.core.post_processing")
    geometry = load("mmdet.core.bbox.geometry", os.path.join(p, "core/bbox/geometry.py"))
    transforms = load("mmdet.core.bbox.transforms", os.path.join(p, "core/bbox/transforms.py"))
    misc = load("mmdet.core.utils.misc", os.path.join(p, "core/utils/misc.py"))
    ops = stub("mmdet.ops")
    stub("mmdet.ops.nms", nms_wrapper=stub("mmdet.ops.nms.nms_wrapper", nms=_nms_stub))
    bbox_nms = load("mmdet.core.post_processing.bbox_nms", os.path.join(p, "core/post_processing/bbox_nms.py"))
    core = sys.modules["mmdet.core"]
    core.distance2bbox, core.bbox_overlaps = transforms.distance2bbox, geometry.bbox_overlaps
    core.bbox2result = transforms.bbox2result
    core.force_fp32, core.auto_fp16, core.multi_apply = _force_fp32, _force_fp32, misc.multi_apply
    core.multiclass_nms = bbox_nms.multiclass_nms
    if hasattr(bbox_nms, "multiclass_nms_idx"):
        core.multiclass_nms_idx = bbox_nms.multiclass_nms_idx
    # mmdet.ops: the pure-Python layers run as written, the compiled ones are stand-ins
    stub("mmdet.ops.dcn", DeformConvPack=_DeformConvPack, ModulatedDeformConvPack=None, DeformConv=_DeformConv)
    for leaf in ("activation", "conv_ws", "norm", "scale"):
        if os.path.exists(os.path.join(p, "ops", leaf + ".py")):
            load("mmdet.ops." + leaf, os.path.join(p, "ops", leaf + ".py"))
    for leaf in ("conv", "conv_module"):
        if os.path.exists(os.path.join(p, "ops", leaf + ".py")):
            load("mmdet.ops." + leaf, os.path.join(p, "ops", leaf + ".py"))
    if "mmdet.ops.conv" in sys.modules:
        ops.build_conv_layer = sys.modules["mmdet.ops.conv"].build_conv_layer
        ops.build_norm_layer = sys.modules["mmdet.ops.norm"].build_norm_layer
        ops.ContextBlock = ops.GeneralizedAttention = None
    if "mmdet.ops.conv_module" in sys.modules:
        ops.ConvModule = sys.modules["mmdet.ops.conv_module"].ConvModule
    if "mmdet.ops.scale" in sys.modules:
        ops.Scale = sys.modules["mmdet.ops.scale"].Scale
    ops.DeformConv, ops.CropSplit, ops.CropSplitGt = _DeformConv, _CropSplit, _CropSplitGt
    # losses
    models = stub("mmdet.models")
    reg = stub("mmdet.models.registry", HEADS=_Registry(), LOSSES=_Registry(), BACKBONES=_Registry(), NECKS=_Registry(),
               DETECTORS=_Registry(), SHARED_HEADS=_Registry(), ROI_EXTRACTORS=_Registry())
    stub("mmdet.models.losses")
    load("mmdet.models.losses.utils", os.path.join(p, "models/losses/utils.py"))
    ops.sigmoid_focal_loss = None       # patched below, once py_sigmoid_focal_loss exists
    focal = load("mmdet.models.losses.focal_loss", os.path.join(p, "models/losses/focal_loss.py"))

    def _sfl(pred, target, gamma, alpha):
        # the CUDA op takes integer labels (0 = background, k = column k-1); the python formula a one-hot target
        onehot = torch.zeros_like(pred)
        idx = target.nonzero().view(-1)
        onehot[idx, target[idx] - 1] = 1
        return focal.py_sigmoid_focal_loss(pred, onehot, gamma=gamma, alpha=alpha, reduction="none")

    focal._sigmoid_focal_loss = _sfl
    iou = load("mmdet.models.losses.iou_loss", os.path.join(p, "models/losses/iou_loss.py"))
    ce = load("mmdet.models.losses.cross_entropy_loss", os.path.join(p, "models/losses/cross_entropy_loss.py"))
    load("mmdet.models.losses.mse_loss", os.path.join(p, "models/losses/mse_loss.py"))

    def build_loss(cfg):
        cfg = dict(cfg)
        return reg.LOSSES.module_dict[cfg.pop("type")](**cfg)

    stub("mmdet.models.builder", build_loss=build_loss)
    wi = load("mmdet.models.utils_weight_init", os.path.join(p, "models/utils/weight_init.py"))
    stub("mmdet.models.utils", bias_init_with_prob=wi.bias_init_with_prob, ConvModule=getattr(ops, "ConvModule", None),
         Scale=getattr(ops, "Scale", None), build_norm_layer=None, build_conv_layer=None)
    stub("mmdet.models.anchor_heads")
    head = load("mmdet.models.anchor_heads.sipmask_head", os.path.join(p, "models/anchor_heads/sipmask_head.py"))
    # backbone and neck (M/ only): ResNet needs mmcv.runner.load_checkpoint and mmdet.utils.get_root_logger at import time
    stub("mmcv.runner", load_checkpoint=None)
    stub("mmdet.utils", get_root_logger=None)
    stub("mmdet.models.backbones")
    stub("mmdet.models.necks")
    resnet = load("mmdet.models.backbones.resnet", os.path.join(p, "models/backbones/resnet.py"))
    fpn = load("mmdet.models.necks.fpn", os.path.join(p, "models/necks/fpn.py"))
    return types.SimpleNamespace(geometry=geometry, transforms=transforms, bbox_nms=bbox_nms, focal=focal, iou=iou, ce=ce,
                                 head=head, build_loss=build_loss, resnet=resnet, fpn=fpn, tag=tag)


def mmdet_tree_vis(root=V):
    """The SipMask-VIS tree (older mmdet layout: ConvModule / Scale live in mmdet/models/utils, the head imports
    `cross_entropy` and `accuracy` from ..losses).  Same stand-ins as mmdet_tree(); additionally
    torch.cuda.current_device() answers "cpu" (the VIS head builds two dummy tensors on `torch.cuda.current_device()`)."""
    for k in [k for k in sys.modules if k == "mmdet" or k.startswith("mmdet.") or k.startswith("mmcv") or
              k.startswith("pycocotools")]:
        del sys.modules[k]
    _mmcv()
    torch.cuda.current_device = lambda: "cpu"
    STAND_INS["torch.cuda.current_device"] = "no GPU in the build container: answers 'cpu' (V/ sipmask_head.py:548-552,626)"
    stub("pycocotools", mask=stub("pycocotools.mask", encode=lambda a: [a]))
    p = os.path.join(root, "mmdet")
    stub("mmdet")
    stub("mmdet.core")
    stub("mmdet.core.bbox")
    stub("mmdet.core.utils")
    stub("mmdet.core.post_processing")
    geometry = load("mmdet.core.bbox.geometry", os.path.join(p, "core/bbox/geometry.py"))
    transforms = load("mmdet.core.bbox.transforms", os.path.join(p, "core/bbox/transforms.py"))
    misc = load("mmdet.core.utils.misc", os.path.join(p, "core/utils/misc.py"))
    ops = stub("mmdet.ops", DeformConv=_DeformConv, CropSplit=_CropSplit, CropSplitGt=_CropSplitGt, sigmoid_focal_loss=None)
    stub("mmdet.ops.nms", nms_wrapper=stub("mmdet.ops.nms.nms_wrapper", nms=_nms_stub))
    bbox_nms = load("mmdet.core.post_processing.bbox_nms", os.path.join(p, "core/post_processing/bbox_nms.py"))
    core = sys.modules["mmdet.core"]
    core.distance2bbox, core.bbox_overlaps = transforms.distance2bbox, geometry.bbox_overlaps
    core.force_fp32, core.auto_fp16, core.multi_apply = _force_fp32, _force_fp32, misc.multi_apply
    core.multiclass_nms, core.multiclass_nms_idx = bbox_nms.multiclass_nms, bbox_nms.multiclass_nms_idx
    stub("mmdet.models")
    reg = stub("mmdet.models.registry", HEADS=_Registry(), LOSSES=_Registry())
    utils = stub("mmdet.models.utils")
    for leaf in ("conv_ws", "norm", "scale", "weight_init", "conv_module"):
        load("mmdet.models.utils." + leaf, os.path.join(p, "models/utils", leaf + ".py"))
    utils.ConvModule = sys.modules["mmdet.models.utils.conv_module"].ConvModule
    utils.Scale = sys.modules["mmdet.models.utils.scale"].Scale
    utils.bias_init_with_prob = sys.modules["mmdet.models.utils.weight_init"].bias_init_with_prob
    losses = stub("mmdet.models.losses")
    load("mmdet.models.losses.utils", os.path.join(p, "models/losses/utils.py"))
    focal = load("mmdet.models.losses.focal_loss", os.path.join(p, "models/losses/focal_loss.py"))

    def _sfl(pred, target, gamma, alpha):
        onehot = torch.zeros_like(pred)
        idx = target.nonzero().view(-1)
        onehot[idx, target[idx] - 1] = 1
        return focal.py_sigmoid_focal_loss(pred, onehot, gamma=gamma, alpha=alpha, reduction="none")

    focal._sigmoid_focal_loss = _sfl
    load("mmdet.models.losses.iou_loss", os.path.join(p, "models/losses/iou_loss.py"))
    ce = load("mmdet.models.losses.cross_entropy_loss", os.path.join(p, "models/losses/cross_entropy_loss.py"))
    acc = load("mmdet.models.losses.accuracy", os.path.join(p, "models/losses/accuracy.py"))
    losses.cross_entropy, losses.accuracy = ce.cross_entropy, acc.accuracy

    def build_loss(cfg):
        cfg = dict(cfg)
        return reg.LOSSES.module_dict[cfg.pop("type")](**cfg)

    stub("mmdet.models.builder", build_loss=build_loss)
    stub("mmdet.models.anchor_heads")
    head = load("mmdet.models.anchor_heads.sipmask_head", os.path.join(p, "models/anchor_heads/sipmask_head.py"))
    return types.SimpleNamespace(geometry=geometry, transforms=transforms, bbox_nms=bbox_nms, head=head, tag="V")


class _DeformConvB(nn.Module):
    """constructor signature of B/fcos_core/layers/dcn/deform_conv_module.py:DeformConv (bias supported there)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert groups == 1
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.stride, self.padding, self.dilation, self.dg = stride, padding, dilation, deformable_groups

    def forward(self, x, offset):
        y = O.deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.dg)
        return y if self.bias is None else y + self.bias.view(1, -1, 1, 1)


def _ml_nms_stub(boxes, scores, labels, thr):
    """B/fcos_core/csrc/cuda/ml_nms.cu through _C.ml_nms -> kept indices (increasing)"""
    from oracle import fcos_core as OB
    keep = OB.ml_nms(boxes.detach().numpy().astype(np.float32), scores.detach().numpy().astype(np.float32),
                     labels.detach().numpy().astype(np.int64), thr)
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def _ns(**kw):
    return types.SimpleNamespace(**kw)


def fcos_cfg(num_classes=81, **over):
    """cfg.MODEL.SIPMASK.* / cfg.TEST.* with the values of B/fcos_core/config/defaults.py:292-314 overridden by
    B/configs/sipmask/sipmask_R_50_FPN_1x.yaml:13-22"""
    sip = dict(NUM_CLASSES=num_classes, FPN_STRIDES=[8, 16, 32, 64, 128], PRIOR_PROB=0.01, INFERENCE_TH=0.05, NMS_TH=0.6,
               PRE_NMS_TOP_N=1000, LOSS_ALPHA=0.25, LOSS_GAMMA=2.0, NUM_CONVS=4, CENTER_SAMPLING_RADIUS=1.5,
               IOU_LOSS_TYPE="giou", NORM_REG_TARGETS=True, CENTERNESS_ON_REG=True, USE_DCN_IN_TOWER=False)
    sip.update(over)
    return _ns(MODEL=_ns(SIPMASK=_ns(**sip)), TEST=_ns(DETECTIONS_PER_IMG=100, BBOX_AUG=_ns(ENABLED=False)))


def fcos_tree(root=B):
    """The maskrcnn-benchmark variant (B/fcos_core).  Stand-ins: _C.nms / _C.ml_nms (compiled) -> oracle NMS per label,
    DeformConv / CropSplit / CropSplitGt (compiled) -> oracle ops; everything else is the reference's Python."""
    for k in [k for k in sys.modules if k == "fcos_core" or k.startswith("fcos_core.") or k.startswith("pycocotools")]:
        del sys.modules[k]
    STAND_INS["fcos_core._C.ml_nms"] = ("compiled extension: replaced by oracle.fcos_core.ml_nms = the golden-pinned greedy "
                                        "NMS run per label (tests/test_targets.py::test_oracle_ml_nms_equals_per_label_nms)")
    stub("pycocotools", mask=stub("pycocotools.mask", encode=lambda a: [a]))
    p = os.path.join(root, "fcos_core")
    stub("fcos_core", _C=_ns(nms=lambda d, s, t: _nms_stub(torch.cat([d, s[:, None]], 1), t)[1], ml_nms=_ml_nms_stub))
    sys.modules["fcos_core._C"] = sys.modules["fcos_core"]._C
    layers = stub("fcos_core.layers", nms=sys.modules["fcos_core"]._C.nms, ml_nms=_ml_nms_stub, DeformConv=_DeformConvB,
                  CropSplit=_CropSplit, CropSplitGt=_CropSplitGt, DFConv2d=None)
    layers.Scale = load("fcos_core.layers.scale", os.path.join(p, "layers/scale.py")).Scale
    layers.IOULoss = load("fcos_core.layers.iou_loss", os.path.join(p, "layers/iou_loss.py")).IOULoss
    layers.SigmoidFocalLoss = load("fcos_core.layers.sigmoid_focal_loss", os.path.join(p, "layers/sigmoid_focal_loss.py")).SigmoidFocalLoss
    stub("fcos_core.structures")
    load("fcos_core.structures.bounding_box", os.path.join(p, "structures/bounding_box.py"))
    load("fcos_core.structures.boxlist_ops", os.path.join(p, "structures/boxlist_ops.py"))
    stub("fcos_core.modeling")
    load("fcos_core.modeling.box_coder", os.path.join(p, "modeling/box_coder.py"))
    load("fcos_core.modeling.utils", os.path.join(p, "modeling/utils.py"))
    load("fcos_core.modeling.matcher", os.path.join(p, "modeling/matcher.py"))
    stub("fcos_core.modeling.rpn")
    load("fcos_core.modeling.rpn.utils", os.path.join(p, "modeling/rpn/utils.py"))
    load("fcos_core.modeling.rpn.inference", os.path.join(p, "modeling/rpn/inference.py"))
    stub("fcos_core.modeling.rpn.sipmask")
    inf = load("fcos_core.modeling.rpn.sipmask.inference", os.path.join(p, "modeling/rpn/sipmask/inference.py"))
    loss = load("fcos_core.modeling.rpn.sipmask.loss", os.path.join(p, "modeling/rpn/sipmask/loss.py"))
    sip = load("fcos_core.modeling.rpn.sipmask.sipmask", os.path.join(p, "modeling/rpn/sipmask/sipmask.py"))
    return types.SimpleNamespace(inference=inf, loss=loss, sipmask=sip, tag="B",
                                 BoxList=sys.modules["fcos_core.structures.bounding_box"].BoxList)
