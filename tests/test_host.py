"""CPU-side tests: plugin seam, parameter names, C-ABI export table (no GPU compute)."""
import ctypes
import os
import re

import pytest
import torch

from oracle import model as OM


def test_library_exports_every_declared_symbol():
    from sipmask_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "sipmask_hip.h")).read()
    declared = set(re.findall(r"\b(sm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsipmask_hip.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.sm_version.restype = ctypes.c_int
    assert lib.sm_version() == 1
    lib.sm_conv_cout_tile.restype = ctypes.c_int
    assert [lib.sm_conv_cout_tile(c) for c in (5, 32, 33, 64, 65, 208, 2048)] == [32, 32, 64, 64, 128, 128, 128]


def test_struct_layout_matches_header():
    """ctypes mirrors of sm_conv_desc / sm_det_desc must have the C sizes (gcc, same ABI)."""
    import subprocess, tempfile
    from sipmask_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = '#include <stdio.h>\n#include "sipmask_hip.h"\nint main(){printf("%zu %zu\\n", sizeof(sm_conv_desc), sizeof(sm_det_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        a, b = subprocess.check_output([os.path.join(d, "t")]).split()
    assert int(a) == ctypes.sizeof(_lib.ConvDesc) and int(b) == ctypes.sizeof(_lib.DetDesc)


def test_registry_contract():
    from sipmask_amd.registry import Registry, build_from_cfg
    R = Registry("thing")

    @R.register_module
    class A:
        def __init__(self, x, y=2):
            self.x, self.y = x, y

    with pytest.raises(KeyError):
        R.register_module(A)
    R.register_module(A, force=True)
    with pytest.raises(TypeError):
        R.register_module(3)
    a = build_from_cfg(dict(type="A", x=1), R, dict(y=5))
    assert (a.x, a.y) == (1, 5)
    assert build_from_cfg(dict(type=A, x=7), R).x == 7
    with pytest.raises(KeyError):
        build_from_cfg(dict(type="B"), R)


def test_detector_builds_from_reference_cfg_and_keys_match_oracle():
    from sipmask_amd.registry import DETECTORS, HEADS, BACKBONES, NECKS, LOSSES
    from sipmask_amd.synthetic import build_synthetic_detector
    det = build_synthetic_detector(50, 0)
    for reg, name in ((DETECTORS, "SipMask"), (HEADS, "SipMaskHead"), (BACKBONES, "ResNet"), (NECKS, "FPN"),
                      (LOSSES, "FocalLoss"), (LOSSES, "IoULoss"), (LOSSES, "CrossEntropyLoss"), (LOSSES, "MSELoss")):
        assert reg.get(name) is not None, name
    sd = det.state_dict()
    ref = OM.init_state_dict(50, 0)
    assert set(sd) == set(ref), sorted(set(sd) ^ set(ref))[:10]
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    # frozen stem + layer1 (frozen_stages=1), BN never trains (requires_grad=False)
    assert not det.backbone.conv1.weight.requires_grad
    assert not det.backbone.layer1[0].conv1.weight.requires_grad
    assert det.backbone.layer2[0].conv1.weight.requires_grad
    assert not det.backbone.layer2[0].bn1.weight.requires_grad
    h = det.bbox_head
    assert len(h.cls_convs) == 3 and len(h.reg_convs) == 4
    assert tuple(h.feat_align.conv_offset.weight.shape) == (72, 4, 1, 1)
    assert h.feat_align.conv_adaption.weight.shape == (256, 256, 3, 3)


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "sipmask_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn


def test_ops_fail_loudly_without_gpu():
    from sipmask_amd import ops as P
    with pytest.raises(NotImplementedError):
        P.nms(torch.rand(4, 5), 0.5)
    with pytest.raises(NotImplementedError):
        P.sigmoid_focal_loss(torch.rand(4, 3), torch.zeros(4, dtype=torch.long))
    dc = P.DeformConv(16, 16, 3, padding=1, deformable_groups=2)
    with pytest.raises(NotImplementedError):
        dc(torch.rand(1, 16, 5, 5), torch.zeros(1, 36, 5, 5))
    if not torch.cuda.is_available():
        from sipmask_amd.engine import SipMaskEngine
        with pytest.raises(RuntimeError):
            SipMaskEngine(OM.init_state_dict(50, 0), 1, (64, 64))
