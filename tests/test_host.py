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


def test_ctypes_prototypes_match_header_arities():
    """every entry point: the number (and pointer/scalar kind) of arguments bound in _lib.PROTOTYPES equals the header's"""
    from sipmask_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "sipmask_hip.h")).read(), flags=re.S)
    decls = list(re.finditer(r"\b(sm_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S))
    assert len(decls) == len(_lib.PROTOTYPES)
    for m in decls:
        name, args = m.group(1), m.group(2).strip()
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        bound = _lib.PROTOTYPES[name][1]
        assert len(params) == len(bound), (name, len(params), len(bound))
        for prm, ct in zip(params, bound):
            is_ptr = "*" in prm or prm.split()[0] == "sm_stream_t"
            ct_ptr = ct is ctypes.c_void_p or ct is ctypes.c_char_p or hasattr(ct, "contents")
            assert is_ptr == ct_ptr, (name, prm, ct)


def test_struct_layout_matches_header():
    """ctypes mirrors of sm_conv_desc / sm_det_desc must have the C sizes (gcc, same ABI)."""
    import subprocess, tempfile
    from sipmask_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = '#include <stdio.h>\n#include "sipmask_hip.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(sm_conv_desc), sizeof(sm_det_desc), sizeof(sm_conv_plan));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        a, b, c = subprocess.check_output([os.path.join(d, "t")]).split()
    assert int(a) == ctypes.sizeof(_lib.ConvDesc) and int(b) == ctypes.sizeof(_lib.DetDesc)
    assert int(c) == ctypes.sizeof(_lib.ConvPlan)


def _plan(sizes, cin, cout, k, stride=1, pad=None, flags=0, B=4, gn=False, deform=False, out_f32=False, res=False):
    from sipmask_amd import hip_ops as H, _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsipmask_hip.so not built (run __graft_entry__.build())")
    pad = k // 2 if pad is None else pad
    lib = _lib.load()
    tile = lib.sm_conv_cout_tile(cout)
    cout_pad = (cout + tile - 1) // tile * tile
    osz = [((h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1) for h, w in sizes]
    lv, lo = H.Levels(B, sizes), H.Levels(B, osz)
    fl = flags | (_lib.SM_CONV_OUT_F32 if out_f32 else 0) | (_lib.SM_CONV_RES_ADD if res else 0)
    d = H.make_conv_desc(B, sizes, osz, lv.row0, lo.row0, cin, cout, cout_pad, k, stride, pad, cin, cout, flags=fl,
                         res_cstride=cout if res else 0, deform_groups=4 if deform else 0)
    return H.conv_plan(d, deformable=deform, with_gn_stats=gn)


PYR = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]          # batch 4 @ 800x1344


def test_conv_launch_plan_rules():
    """sm_conv_plan_query: the launcher's selection rules on the shapes of the R50 plan.
    Host logic only -- no GPU involved."""
    # tower conv with fused GN statistics: 128x128 tiles, 64-wide K, pipelined K loop, 1404 blocks
    p = _plan(PYR, 256, 256, 3, gn=True)
    assert (p["lds_dma"], p["k_step"], p["tile_cout"], p["tile_pos"], p["threads"], p["k_loop"]) == (1, 64, 128, 128, 256, 3)
    assert p["blocks"] == 1404 and p["k_padded"] == 2304
    # K <= 1152 -> 32-wide K steps (but never with fused GN statistics)
    p = _plan([(200, 336)], 64, 64, 3)
    assert (p["k_step"], p["tile_cout"], p["k_loop"], p["k_padded"]) == (32, 64, 0, 576)
    assert _plan([(200, 336)], 64, 64, 3, gn=True)["k_step"] == 64
    p = _plan([(800, 1344)], 8, 64, 7, stride=2, pad=3)                           # stem: cin 3 padded to 8, K 392 -> 448
    assert (p["k_step"], p["k_padded"], p["k_loop"]) == (32, 448, 0)
    # small-M layers shrink the tiles until the launch has >= 512 blocks (or run out of candidates)
    p = _plan([(25, 42)], 512, 512, 3, flags=0x00010000)                          # layer4 3x3: M = 4200 (split-K off)
    assert (p["k_step"], p["tile_cout"], p["tile_pos"]) == (64, 64, 64) and p["blocks"] == 66 * 8 and p["split_k"] == 1
    # ... unless a workspace is offered (sm_conv2d_ws): <= 160 tiles with >= 36 K steps keep the large tile and are cut
    # into K slices instead (f32 partial slabs + reduce kernel)
    p = _plan([(25, 42)], 512, 512, 3)
    assert (p["tile_cout"], p["tile_pos"], p["blocks"], p["split_k"]) == (128, 128, 33 * 4, 4)
    assert p["workspace_bytes"] == 4 * 4200 * 512 * 4
    assert _plan([(25, 42)], 2048, 512, 1)["split_k"] == 1                        # 1x1, 32 K steps: measured slower split
    p = _plan([(50, 84)], 256, 256, 3)                                            # layer3 3x3: M = 16800
    assert (p["tile_cout"], p["tile_pos"]) == (128, 64) and p["blocks"] == 263 * 2 and p["k_loop"] == 3
    # operands VALU must touch are register-staged: deformable gather, input ReLU
    p = _plan(PYR, 256, 256, 3, deform=True)                                       # 256-cout x 128-position tile on 8 waves
    assert (p["lds_dma"], p["tile_cout"], p["tile_pos"], p["k_loop"], p["threads"]) == (0, 256, 128, 0, 512)
    assert _plan([(13, 21)], 256, 256, 3, stride=2, flags=16)["lds_dma"] == 0    # SM_CONV_IN_RELU (P7)
    # the rejected A/B variants of rounds 1-2 (legacy / flat loops, 128x256 tiles, warp specialisation ...) are not part of
    # the default build (csrc/experiments.h, `make EXPERIMENTS=1`): their former flag bits select nothing
    for dead in (0x00100000, 0x00200000, 0x00800000, 0x02000000, 0x00080000):
        q = _plan(PYR, 256, 256, 3, flags=dead)
        assert (q["k_loop"], q["tile_cout"], q["tile_pos"], q["warp_spec"]) == (3, 128, 128, 0), hex(dead)
    # plan selectors that ARE part of the interface: forced K widths
    assert _plan(PYR, 256, 256, 3, flags=0x10000000)["k_step"] == 32
    assert _plan([(200, 336)], 64, 64, 3, flags=0x08000000)["k_step"] == 64
    # 256x256 8-wave tiles (opt-in): only where the rounds of 256 blocks are >= 65 % filled
    p = _plan(PYR, 256, 256, 3, flags=0x00400000, gn=True)
    assert (p["tile_cout"], p["tile_pos"], p["threads"], p["k_loop"], p["blocks"]) == (256, 256, 512, 3, 353)
    p = _plan(PYR[:1], 256, 256, 3, flags=0x00400000)                             # fpn.out0: 263 blocks = 2 rounds, 51 %
    assert (p["tile_cout"], p["tile_pos"], p["threads"]) == (128, 128, 256)
    assert _plan(PYR[:1], 256, 256, 3, flags=0x04400000)["tile_cout"] == 256     # ... unless forced (BIG_TILES)
    assert _plan(PYR, 256, 208, 3, flags=0x00400000, out_f32=True)["tile_cout"] == 256   # cls+cof: cout_pad 256
    assert _plan(PYR, 256, 4, 3, flags=0x00400000)["tile_cout"] == 32            # 32-cout family untouched
    # binary16 operands (SM_CONV_F16, the x3 head plan): the same plan as the bf16 launch of that shape, incl. fused GN
    # statistics with f32 output
    a = _plan(PYR, 768, 256, 3, flags=0x00020000, gn=True, out_f32=True)
    b = _plan(PYR, 768, 256, 3, gn=True)
    assert a == b and a["k_padded"] == 9 * 768


def test_conv_launch_plan_rejects_bad_descriptors():
    from sipmask_amd import hip_ops as H, _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsipmask_hip.so not built (run __graft_entry__.build())")
    lv = H.Levels(2, [(10, 12)])
    ok = dict(batch=2, in_sizes=[(10, 12)], out_sizes=[(10, 12)], in_row0=lv.row0, out_row0=lv.row0, cin=64, cout=128,
              cout_pad=128, k=3, stride=1, pad=1, in_cstride=64, out_cstride=128)
    assert H.conv_plan(H.make_conv_desc(**ok))["blocks"] > 0
    for bad in (dict(cin=60), dict(out_sizes=[(9, 12)]), dict(cout_pad=96), dict(in_cstride=60), dict(batch=0)):
        with pytest.raises(RuntimeError):
            H.conv_plan(H.make_conv_desc(**dict(ok, **bad)))
    # fused GN statistics go with f32 output as well (the x3 plan normalises f32 rows)
    assert H.conv_plan(H.make_conv_desc(**dict(ok, flags=_lib.SM_CONV_OUT_F32)), with_gn_stats=True)["blocks"] > 0
    # round 3: deformable convs take any stride (the offset rows are output rows); what stays rejected is a group count
    # that does not divide the channels into multiples of 8
    assert H.conv_plan(H.make_conv_desc(**dict(ok, stride=2, out_sizes=[(5, 6)], deform_groups=1)), deformable=True)["blocks"] > 0
    with pytest.raises(RuntimeError):
        H.conv_plan(H.make_conv_desc(**dict(ok, deform_groups=3)), deformable=True)


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


@pytest.mark.parametrize("variant,nconv", [("r50", 73), ("ssd", 69), ("vis", 74), ("benchmark", 73), ("dcn", 74)])
def test_engine_plan_builds_without_gpu_and_every_launch_has_a_plan(variant, nconv):
    """The static launch plan is host logic: tools/plan_dump.py builds the engine of every front-end variant on the CPU
    (nothing is launched) and asks sm_conv_plan_query for each prepared conv descriptor."""
    import importlib.util
    from sipmask_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsipmask_hip.so not built (run __graft_entry__.build())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("plan_dump", os.path.join(root, "tools", "plan_dump.py"))
    pd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pd)
    avail = torch.cuda.is_available
    eng = pd.build_on_cpu(variant, batch=2, hw=(256, 320))
    assert torch.cuda.is_available is avail                       # the patch is undone
    allrows = pd.conv_rows(eng)
    rows = {r["name"]: r for r in allrows if r["kind"] != "fused"}          # one row per conv launch ...
    assert sum(1 for r in allrows if r["kind"] == "fused") == len(eng.fused)   # ... and one per fused bottleneck launch
    from sipmask_amd import engine as E
    # the cls and reg tower convs of one depth run as ONE grouped launch:
    # every "head.towerN" launch stands for two of the nconv convolutions
    ngrouped = sum(1 for n in rows if n.startswith("head.tower"))
    grouped = ngrouped > 0
    assert grouped == (E._GROUPED_TOWERS and eng.flag_norm)   # GN heads only
    # fused bottleneck tails (layer1 / layer2, plain conv2): conv2 + conv3 (+ the next block's conv1) per launch
    # (round 4: layer1's first tail also carries the block's 1x1 shortcut conv -- three convs in that launch)
    pairs = [t for t in eng.fused if not hasattr(t, "w2")]               # layer3: conv3 + next conv1 (two 1x1 convs per launch)
    tails = [t for t in eng.fused if hasattr(t, "w2")]
    nfused = sum(2 + (t.w1n is not None) + (t.x_block is not None) for t in tails) + 2 * len(pairs)
    nshort = sum(1 for t in tails if t.x_block is not None)
    if variant in ("r50", "vis", "benchmark", "ssd") and E._FUSE_BOTTLENECK == 1:
        nchain = sum(1 for t in tails if t.w1n is not None)              # layer1: the next block's conv1 rides along
        assert nchain == (2 if E._CHAIN_CONV1 else 0)
        assert len(tails) == 7 and nfused == 7 * 2 + nshort + nchain + 2 * len(pairs)
        assert len(pairs) == (5 if E._PAIR_1X1 else 0)       # R50 layer3: 6 blocks (off)
        assert nshort == E._FUSE_SHORTCUT
        assert ("backbone.layer1.0.downsample" in rows) == (nshort == 0) and ("backbone.layer2.0.downsample" in rows) == (nshort < 2)
    # sip_mask_lat0 by linearity (round 4): the 768 -> 512 conv runs as three 1x1 convs (l0, l1, l2) + sm_upsample_sum2
    nlin = 2 if getattr(eng, "lat0_by_linearity", False) else 0
    if nlin:
        assert {"head.sip_mask_lat0", "head.sip_mask_lat0.l1", "head.sip_mask_lat0.l2"} <= set(rows)
        assert any(lbl == "up:sum2" for lbl, _ in eng.steps) and not any(lbl.startswith("up:cat") for lbl, _ in eng.steps)
    # the stem (conv1 + bn1 + relu + maxpool) is one launch of its own kernel (round 4, csrc/stem_fused.hip), not a conv row
    nstem = 1 if any(lbl == "stem_fused" for lbl, _ in eng.steps) else 0
    assert nstem == (1 if E._STEM_FUSED else 0)
    assert not (nstem and any(lbl in ("nhwc", "maxpool", "conv:stem") for lbl, _ in eng.steps))
    assert len(rows) == len(eng.convs) and len(rows) == nconv - ngrouped - nfused + nlin - nstem
    assert all(r["blocks"] > 0 and r["waves"] > 0 and r["kind"] in ("igemm", "patch", "window", "smallco") for r in rows.values())
    assert len(eng.steps) == len(eng.lanes)
    joined = set()
    for lane in eng.lanes:                                           # every side lane that is used gets joined
        if isinstance(lane, tuple):
            joined.update(lane[1:])
    assert {l for l in eng.lanes if isinstance(l, int) and l > 0} <= joined
    # every descriptor stays inside the buffers it was prepared for (rows per level, channel strides, weight matrix)
    for c in eng.convs:
        d = c.desc
        K = d.kh * d.kw * d.cin
        G = max(int(d.ngroups), 1)
        if getattr(c, "smallco", False):                              # MFMA-fragment order: [cin / 32][9][2][64 lanes][8]
            assert tuple(c.w.shape) == (d.cin // 32, 9, 2, 64, 8) and d.cout_pad == 32 and d.cout <= 32, c.name
        else:
            assert tuple(c.w.shape)[-2:] == (d.cout_pad, (K + 63) // 64 * 64) and c.w.numel() == G * d.cout_pad * ((K + 63) // 64 * 64), c.name
        if G > 1:
            assert d.w_group_stride == d.cout_pad * ((K + 63) // 64 * 64) and c.gn_stats.numel() == G * d.gn_group_stride, c.name
        assert c.x.dim() == 2 and c.y.dim() == 2 and d.in_cstride <= c.x.shape[1] and d.cin <= d.in_cstride, c.name
        assert d.out_coff + d.cout <= d.out_cstride <= c.y.shape[1], c.name
        for l in range(d.nlev):
            assert (G - 1) * d.x_group_rows + d.in_row0[l] + d.batch * d.in_h[l] * d.in_w[l] <= c.x.shape[0], (c.name, l)
            assert (G - 1) * d.y_group_rows + d.out_row0[l] + d.batch * d.out_h[l] * d.out_w[l] <= c.y.shape[0], (c.name, l)
            if c.residual is not None and (d.flags & 8):            # SM_CONV_RES_NEAREST
                assert d.res_row0[l] + d.batch * d.res_h[l] * d.res_w[l] <= c.residual.shape[0], (c.name, l)
            elif c.residual is not None:
                assert d.out_row0[l] + d.batch * d.out_h[l] * d.out_w[l] <= c.residual.shape[0] and \
                    d.res_cstride <= c.residual.shape[1], (c.name, l)
        if c.offset is not None:
            rows_out = sum(d.batch * d.out_h[l] * d.out_w[l] for l in range(d.nlev))
            assert c.offset.numel() >= rows_out * d.deform_groups * d.kh * d.kw * 2, c.name
        if c.bias is not None:
            assert c.bias.numel() >= d.cout, c.name
    # FeatureAlign (3x3, 4 deformable groups of 64 channels): the LDS-window kernel, 8 x 32-position tiles per image and level
    fa = rows["head.feat_align"]
    assert fa["kind"] == "window" and fa["shape"] == "256x(8x32)"
    def fa_tiles(h, w):            # row tiles 8 x 32; a level's right-hand strip as 32 x 8 column tiles where they are fewer
        nty, nfull, rem = -(-h // 8), w // 32, w % 32
        strip = -(-rem // 8) * -(-h // 32)
        return nty * nfull + (min(strip, nty) if rem else 0)
    assert fa["blocks"] == 2 * sum(fa_tiles(h, w) for h, w in eng.lv.sizes)
    # P7 = conv(relu(P6)) (fpn.py:166-170): the plan feeds it a ReLU'd copy so it keeps the LDS-DMA path
    assert rows["fpn.p7"]["plan"]["lds_dma"] == 1 and any(lbl == "relu:p6" for lbl, _ in eng.steps)
    tower = rows["head.tower0" if grouped and "head.tower0" in rows else "head.reg_convs.0"]
    assert tower["kind"] == "patch" or (tower["plan"]["k_step"], tower["plan"]["k_loop"]) == (64, 3)
    if variant == "dcn":
        assert sum(1 for n in rows if n.endswith("conv2.conv_offset")) == 5
        assert all(rows[n[:-len(".conv_offset")]]["plan"]["lds_dma"] == 0 for n in rows if n.endswith("conv2.conv_offset"))
    if variant == "vis":
        assert "head.sipmask_track" in rows or any(n.startswith("head.track") for n in rows)


def test_patch_conv_launch_shapes():
    """sm_conv3x3_patch_plan (pure host logic): the launch shapes the planner picks for the BASELINE head launches and
    its invariants -- every shape covers every segment, never estimates worse than the uniform launch, and the A/B
    flag restores the uniform launch."""
    from sipmask_amd import hip_ops as H
    levels = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    expect = {(2, 2, 5): (256, 208, 128), (2, 1, 5): (0, 242, 192), (2, 1, 1): (0, 178, 192), (4, 1, 1): (256, 20, 128),
              (4, 2, 5): (736, 0, 128)}
    for (b, g, nl), (big, small, spos) in expect.items():
        sizes = levels[:nl]
        lv = H.Levels(b, sizes)
        d = H.make_conv_desc(b, sizes, sizes, lv.row0, lv.row0, 256, 256, 256, 3, 1, 1, 256, 256, ngroups=g)
        pl = H.conv3x3_patch_plan(d)
        assert (pl["big"], pl["small"]) == (big, small), (b, g, nl, pl)
        if small:
            assert pl["small_pos"] == spos
        assert H.conv3x3_patch_tiles(d) == big + small
        d.flags = 0x4000
        un = H.conv3x3_patch_plan(d)
        assert un["small"] == 0 and un["makespan"] >= pl["makespan"]
        assert un["big"] == g * b * sum(-(-h * (w + 2) // 256) for h, w in sizes)
    # ragged shapes: positions covered = segment length, for random geometries and every forced variant
    import random
    rnd = random.Random(0)
    for _ in range(50):
        nl = rnd.randint(1, 5)
        sizes = [(rnd.randint(1, 120), rnd.randint(1, 250)) for _ in range(nl)]
        b, g = rnd.randint(1, 6), rnd.randint(1, 2)
        co = rnd.choice([8, 208, 256, 512])
        lv = H.Levels(b, sizes)
        for flag in (0, 0x2000, 0x1000):
            d = H.make_conv_desc(b, sizes, sizes, lv.row0, lv.row0, 64, co, (co + 255) // 256 * 256, 3, 1, 1, 64, co,
                                 flags=flag, ngroups=g)
            pl = H.conv3x3_patch_plan(d)
            segs = b * g * ((co + 255) // 256)
            assert pl["big"] % segs == 0 and pl["small"] % segs == 0
            cover = pl["big"] * 256 + pl["small"] * pl["small_pos"]
            need = segs * sum(h * (w + 2) for h, w in sizes)
            assert cover >= need and cover < need + segs * nl * 256, (sizes, b, g, co, pl)


def test_deform_conv_kernel_choice_is_host_logic():
    """sm_deform_conv_window_plan (no GPU): FeatureAlign's shape -- 3x3, 64 channels per deformable group, 256-cout tiles --
    runs on the LDS-window kernel with 8 x 32-position row tiles (+ 32 x 8 column tiles on a level's right-hand strip where
    they are fewer, round 5); the backbone DCN of SipMask++ (1 deformable group), a 1x1
    kernel, a residual epilogue and the A/B flag stay on the gather loader."""
    from sipmask_amd import hip_ops as H, _lib
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    lv = H.Levels(2, sizes)
    mk = lambda cin, co, k, G, flags=0, szs=sizes, l=lv: H.make_conv_desc(
        2, szs, szs, l.row0, l.row0, cin, co, (co + 255) // 256 * 256, k, 1, k // 2, cin, co, flags=flags, deform_groups=G)
    pl = H.deform_conv_window_plan(mk(256, 256, 3, 4))
    # row tiles ceil(h / 8) * floor(w / 32) + the strip: (13*5 + 4) + (7*2 + 6) + (4*1 + 2) + 2*1 + 1*1 = 98 per image
    # (all row tiles: 13*6 + 7*3 + 4*2 + 2 + 1 = 110)
    assert pl == dict(blocks=2 * 98, tile=(8, 32), window_pixels=16 * 40)
    assert H.deform_conv_window_plan(mk(256, 512, 3, 4))["blocks"] == 2 * 98 * 2           # two 256-cout tiles
    assert H.deform_conv_window_plan(mk(256, 256, 3, 4, flags=_lib.SM_CONV_DBG_DEFORM_GATHER)) is None
    assert H.deform_conv_window_plan(mk(256, 256, 3, 1)) is None                            # 256 channels per group
    assert H.deform_conv_window_plan(mk(128, 256, 3, 4)) is None                            # 32 channels per group
    assert H.deform_conv_window_plan(mk(256, 256, 1, 4)) is None
    assert H.deform_conv_window_plan(mk(256, 256, 3, 4, flags=_lib.SM_CONV_RES_ADD)) is None
    d = mk(256, 256, 3, 4)
    d.cout_pad = 128 * 3                                                                     # not a multiple of 256
    assert H.deform_conv_window_plan(d) is None


def test_deform_conv_argument_checks_run_before_any_launch():
    """ops.deform_conv (M/mmdet/ops/dcn/deform_conv.py:16-58,99): the argument checks of the grouped wrapper are host logic --
    a group count that does not divide the channels is a ValueError, conv groups / deformable groups where neither divides the
    other a NotImplementedError, and a CPU tensor reaches the reference's NotImplementedError (no CPU implementation)."""
    import torch
    from sipmask_amd import ops as P
    x = torch.zeros(1, 12, 5, 5)
    w = torch.zeros(8, 4, 3, 3)
    with pytest.raises(ValueError):
        P.deform_conv(x, torch.zeros(1, 18, 5, 5), torch.zeros(8, 5, 3, 3), 1, 1, 1, 3, 1)       # weight does not fit groups
    with pytest.raises(NotImplementedError):
        P.deform_conv(torch.zeros(1, 24, 5, 5), torch.zeros(1, 72, 5, 5), torch.zeros(12, 4, 3, 3), 1, 1, 1, 6, 4)
    with pytest.raises(NotImplementedError):
        P.deform_conv(x, torch.zeros(1, 18, 5, 5), torch.zeros(8, 12, 3, 3), 1, 1, 1, 1, 1)      # CPU tensors
    with pytest.raises(ValueError):
        P.deform_conv(torch.zeros(12, 5, 5), torch.zeros(1, 18, 5, 5), w)                         # not 4-D
    m = P.DeformConv(16, 8, (1, 3), stride=2, padding=(0, 1), groups=2, deformable_groups=2)
    assert m.weight.shape == (8, 8, 1, 3) and m.kernel_size == (1, 3) and m.padding == (0, 1) and m.groups == 2


def test_training_rows_pick_the_big_tile_for_wide_pyramid_convs():
    from sipmask_amd import ops_rows as R, _lib
    big = _lib.SM_CONV_DBG_TILE256 | _lib.SM_CONV_DBG_HAND_PLACED
    assert R._big_tile_flags(3, 1, 256, 89600) == big
    assert R._big_tile_flags(3, 1, 256, 20000) == 0          # short position axis: the library's own rule
    assert R._big_tile_flags(1, 1, 256, 89600) == 0 and R._big_tile_flags(3, 2, 256, 89600) == 0
    assert R._big_tile_flags(3, 1, 208, 89600) == 0          # couts that do not fill the 256-wide tile


def test_fcos_sipmask_head_alias_builds_from_cfg():
    """north_star names two registry entries, "SipMaskHead / FCOSSipMaskHead"; the second denotes the FCOS-style head of the
    maskrcnn-benchmark variant (B/fcos_core/modeling/rpn/sipmask/sipmask.py:48-190): an mmdet-style cfg dict
    builds it with B/'s parameter names, and the detector exposes forward_dummy (single_stage.py:52-59)."""
    import sipmask_amd.detector as D
    from sipmask_amd.benchmark_train import SipMaskBenchmarkHead
    from sipmask_amd.registry import HEADS, build_head
    assert {"SipMaskHead", "FCOSSipMaskHead"} <= set(HEADS.module_dict)
    h = build_head(dict(type="FCOSSipMaskHead", num_classes=81, in_channels=256, stacked_convs=4, feat_channels=256,
                        strides=[8, 16, 32, 64, 128]))
    assert isinstance(h, SipMaskBenchmarkHead) and h.fpn_strides == (8, 16, 32, 64, 128)
    keys = set(h.state_dict())
    # B/ names: towers as Sequential(conv, GN, ReLU) x N with conv bias, cls_logits / bbox_pred / centerness, 5 Scales
    assert {"cls_tower.0.weight", "cls_tower.0.bias", "cls_tower.7.weight", "bbox_tower.10.bias", "cls_logits.bias",
            "bbox_pred.weight", "centerness.weight", "scales.4.scale", "feat_align.conv_adaption.bias",
            "sip_cof.weight", "sip_mask_lat0.weight"} <= keys
    assert "cls_tower.9.weight" not in keys and "bbox_tower.9.weight" in keys          # cls tower: one conv fewer
    b = build_head(dict(type="FCOSSipMaskHead", num_convs=2, fpn_strides=(8, 16, 32, 64, 128)))
    assert "bbox_tower.3.weight" in b.state_dict() and "bbox_tower.6.weight" not in b.state_dict()
    import pytest
    with pytest.raises(ValueError):
        build_head(dict(type="FCOSSipMaskHead", num_convs=2, stacked_convs=4))
    with pytest.raises(TypeError):
        build_head(dict(type="FCOSSipMaskHead", loss_cls=dict(type="FocalLoss")))
    assert callable(getattr(D.SipMask, "forward_dummy"))


def test_launch_plan_decisions_at_the_baseline_shape():
    """Round 4's launch-plan rules, pinned on the CPU at BASELINE configs[1] (R50, batch 4, 800 x 1344, a PipelinedPlan slot):
    the FPN's three output convs are ONE patch launch with per-level weights; layer3 / layer4 conv2 take the 128-cout patch
    tile; sip_mask_lat and fcos_reg + centerness their own small-cout kernel; the stem is one launch; sip_mask_lat0 runs by linearity; lat2 / P6 / P7 keep the
    latency-shaped plan (no big-tile flag); the conv FLOPs of a step are the reference's 1 803.7 GFLOP minus what the
    linearity saves."""
    import importlib.util
    from sipmask_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsipmask_hip.so not built (run __graft_entry__.build())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("plan_dump", os.path.join(root, "tools", "plan_dump.py"))
    pd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pd)
    eng = pd.build_on_cpu("r50", batch=4, hw=(800, 1344), pipelined=True)
    convs = {c.name: c for c in eng.convs}
    outs = convs["fpn.outs"]
    assert eng.fpn_grouped and outs.patch and outs.desc.nlev == 3 and outs.desc.w_level_stride == 256 * 9 * 256
    assert [tuple(s) for s in zip(outs.desc.in_h[:3], outs.desc.in_w[:3])] == [(100, 168), (50, 84), (25, 42)]
    assert not any(n.startswith("fpn.out") and n != "fpn.outs" for n in convs)
    for n in ("backbone.layer3.2.conv2", "backbone.layer4.1.conv2"):
        assert convs[n].patch and convs[n].desc.patch_cout_tile == 128, n
    assert convs["head.tower0"].patch and convs["head.tower0"].desc.patch_cout_tile == 0 and convs["head.tower0"].desc.cout_pad == 256
    for n, co in (("head.sip_mask_lat", 32), ("head.reg_ctr", 8)):     # the small-cout kernel (one wave per 2 x 32 tile)
        assert convs[n].smallco and not convs[n].patch and convs[n].desc.cout_pad == 32 and convs[n].desc.cout == co, n
    assert not any(c.smallco for n, c in convs.items() if n not in ("head.sip_mask_lat", "head.reg_ctr"))
    assert any(lbl == "stem_fused" for lbl, _ in eng.steps) and "stem" not in convs
    assert eng.lat0_by_linearity and convs["head.sip_mask_lat0"].desc.cin == 256 and convs["head.sip_mask_lat0"].residual is not None
    big = _lib.SM_CONV_DBG_BIG_TILES
    assert convs["backbone.layer3.2.conv1"].desc.flags & big                 # a slot builds for CU time ...
    for n in ("fpn.lat2", "fpn.p6", "fpn.p7"):
        assert not (convs[n].desc.flags & big), n                              # ... except the three launch-latency-shaped convs
    gf = eng.total_conv_flops() / 1e9
    assert abs(gf - (1803.7 - 29.7)) < 0.5, gf


def test_round4_weight_layouts_of_the_new_kernels():
    """Host-side operand layouts of the round-4 kernels, checked element by element against their definitions (no GPU):
    sm_stem_fused's [64][7][8][4] weights (kw = 7 and cin = 3 slots zero), sm_conv3x3_smallco's MFMA A fragments
    [cin / 32][9][2][64][8] (lane = 32 * khalf + cout row, channel = 32 * slice + 16 * half + 8 * khalf + e), and the
    [w3 | w_downsample] rows + summed bias of the fused-shortcut bottleneck tail."""
    import torch
    from sipmask_amd import hip_ops as H
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 3, 7, 7, generator=g)
    ws = H.prep_stem_weight(w).float()
    assert tuple(ws.shape) == (64, 7, 8, 4)
    assert float(ws[:, :, 7].abs().max()) == 0 and float(ws[..., 3].abs().max()) == 0
    for co, kh, kw, c in ((0, 0, 0, 0), (63, 6, 6, 2), (17, 3, 5, 1)):
        assert float(ws[co, kh, kw, c]) == float(w[co, c, kh, kw].to(torch.bfloat16))
    with pytest.raises(ValueError):
        H.prep_stem_weight(torch.zeros(64, 3, 3, 3))
    co_, ci = 24, 96
    w2 = torch.randn(co_, ci, 3, 3, generator=g)
    wf = H.prep_conv_weight_smallco(w2).float()
    assert tuple(wf.shape) == (ci // 32, 9, 2, 64, 8)
    for (m, ch, kh, kw) in ((0, 0, 0, 0), (23, 95, 2, 2), (5, 40, 1, 2), (11, 63, 0, 1)):
        sl, r = divmod(ch, 32)
        half, r = divmod(r, 16)
        khalf, e = divmod(r, 8)
        assert float(wf[sl, kh * 3 + kw, half, 32 * khalf + m, e]) == float(w2[m, ch, kh, kw].to(torch.bfloat16))
    assert float(wf[:, :, :, 24:32].abs().max()) == 0 and float(wf[:, :, :, 56:64].abs().max()) == 0      # cout rows 24..31: padding
    with pytest.raises(ValueError):
        H.prep_conv_weight_smallco(torch.zeros(40, 64, 3, 3))


def test_deform_x3_tile_plan_host_logic():
    """pure host arithmetic (sm_deform_conv2d_x3_plan): column tiles replace the row tiles of a level's right-hand strip
    when they are fewer -- BASELINE's pyramid (100x168 ... 7x11) is 98 tiles per image instead of 110"""
    from sipmask_amd import hip_ops as H, _lib
    S = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    lv = H.Levels(4, S)
    d = H.make_conv_desc(4, S, S, lv.row0, lv.row0, 256, 256, 256, 3, 1, 1, 256, 256, deform_groups=4,
                         flags=_lib.SM_CONV_F16 | _lib.SM_CONV_OUT_F32)
    p = H.deform_conv2d_x3_plan(d)
    assert p == dict(blocks=392, row_tiles=4 * (65 + 14 + 4 + 2 + 1), col_tiles=4 * (4 + 6 + 2), window_pixels=640)
    SIZES = [(40, 72), (19, 45), (10, 23), (33, 8), (2, 3)]
    lv = H.Levels(2, SIZES)
    d = H.make_conv_desc(2, SIZES, SIZES, lv.row0, lv.row0, 256, 256, 256, 3, 1, 1, 256, 256, deform_groups=4,
                         flags=_lib.SM_CONV_F16 | _lib.SM_CONV_OUT_F32)
    p = H.deform_conv2d_x3_plan(d)
    assert (p["row_tiles"], p["col_tiles"]) == (2 * (10 + 3 + 2 + 0 + 1), 2 * (2 + 2 + 0 + 2 + 0))
    d.cin = 128                                   # 32 channels per deformable group: not this kernel's shape
    assert H.deform_conv2d_x3_plan(d) is None and not H.deform_conv2d_x3_supported(d)


def test_bench_configs_and_synthetic_ssd_detector():
    """bench.py's configurations parse to the batch / steps the contract's defaults promise, and the synthetic detector of
    `--config ssd` is the 544 x 544 SSD-style head (sipmask_r50_caffe_fpn_ssd_6x.py:23-31,54-59: two tower convs, no norm,
    ssd_flag, score_thr 0.1)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = bench.parse([])
    assert (a.config, a.gpus, a.batch, a.depth) == ("r50", 1, 4, 50) and a.steps >= 50 and a.warmup >= 10
    a = bench.parse(["--config", "ssd"])
    assert (a.batch, a.depth, a.steps) == (8, 50, 50) and bench.SSD_HW == (544, 544)
    assert bench.parse(["--config", "r101"]).depth == 101
    from sipmask_amd.synthetic import build_synthetic_detector
    det = build_synthetic_detector(50, ssd=True)
    h = det.bbox_head
    assert h.ssd_flag and h.stacked_convs == 2 and h.norm_cfg is None and det.test_cfg["score_thr"] == 0.1
    assert len(h.reg_convs) == 2 and not any(".gn." in k for k in h.state_dict())
    boxes, info = bench.coco_boxes(2, 2, 100, 544, 544, scale_xy=(544 / 640.0, 544 / 480.0))
    assert boxes.shape == (2, 2, 100, 4) and float(boxes[..., 2].max()) <= 544 and float(boxes[..., 3].max()) <= 544
    assert 0.02 < info["mean_box_share_of_image"] < 0.08
