#!/usr/bin/env python
"""FeatureAlign's deformable conv (3x3, 256 -> 256, 4 deformable groups, 5 FPN levels of the 800x1344 input): the LDS-patch
kernel (deform_patch.hip) against the global-gather loader (SM_CONV_DBG_DEFORM_GATHER), by offset magnitude."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H, _lib
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
dev = torch.device("cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
BS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 4]
SCALES = [float(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.0, 0.3, 1.0, 2.0, 4.0]
for B in BS:
    lv = H.Levels(B, LEVELS)
    x = (torch.randn(lv.rows, 256, device=dev) * 0.5).to(torch.bfloat16)
    w = torch.randn(256, 256, 3, 3, device=dev) / 48
    wq, cp = H.prep_conv_weight(w, 256)
    y = torch.empty(lv.rows, 256, dtype=torch.bfloat16, device=dev)
    st = H.gn_stats_alloc(B * 5 * 32, dev)
    for scale in SCALES:
        off = torch.randn(lv.rows, 72, device=dev) * scale
        res = {}
        for name, fl in (("patch", 0), ("gather", _lib.SM_CONV_DBG_DEFORM_GATHER)):
            d = H.make_conv_desc(B, LEVELS, LEVELS, lv.row0, lv.row0, 256, 256, cp, 3, 1, 1, 256, 256, flags=fl, deform_groups=4)
            ts = []
            for rnd in range(4):
                e0.record()
                for _ in range(10):
                    H.conv2d_gn_stats(d, x, off, wq, None, None, y, st)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    ts.append(e0.elapsed_time(e1) / 10)
            res[name] = sorted(ts)[len(ts) // 2]
        flops = 2.0 * lv.rows * 256 * 2304
        print("B=%d offsets ~N(0,%.1f): " % (B, scale) + "   ".join("%s %.4f ms %.0f TF/s" % (k, v, flops / v / 1e9) for k, v in res.items()))

# ---- the x3 head plan's FeatureAlign (round 5): f32 rows, split-precision contraction.  window = csrc/deform_patch_x3.hip
# gather = conv_f32.hip's loader (the round-3 kernel)
for B in BS:
    lv = H.Levels(B, LEVELS)
    x = (torch.randn(lv.rows, 256, device=dev) * 0.5).abs()
    w = torch.randn(256, 256, 3, 3, device=dev) / 48
    scale = H.x3_weight_scale([w])
    w_win, cp = H.prep_deform_weight_x3(w, scale, 4)
    w_gat, cpg = H.prep_conv_weight_f32(w * scale, 256)
    y = torch.empty(lv.rows, 256, dtype=torch.float32, device=dev)
    st = H.gn_stats_alloc(B * 5 * 32, dev)
    mk = lambda fl, cpad: H.make_conv_desc(B, LEVELS, LEVELS, lv.row0, lv.row0, 256, 256, cpad, 3, 1, 1, 256, 256, flags=fl,
                                           deform_groups=4, acc_scale=1.0 / scale)
    F16, F32O = _lib.SM_CONV_F16, _lib.SM_CONV_OUT_F32
    print("x3 plan:", H.deform_conv2d_x3_plan(mk(F16 | F32O, cp)))
    for scale_o in SCALES:
        off = torch.randn(lv.rows, 72, device=dev) * scale_o
        runs = (("window", lambda d=mk(F16 | F32O, cp): H.deform_conv2d_x3(d, x, off, w_win, None, y, st)),
                ("gather", lambda d=mk(F16, cpg): H.conv2d_f32(d, x, off, w_gat, None, None, y)))
        res = {}
        for name, fn in runs:
            ts = []
            for rnd in range(4):
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    ts.append(e0.elapsed_time(e1) / 10)
            res[name] = sorted(ts)[len(ts) // 2]
        flops = 2.0 * lv.rows * 256 * 2304
        print("x3 B=%d offsets ~N(0,%.1f): " % (B, scale_o) +
              "   ".join("%s %.4f ms %.0f TF/s (x3 MFMA work %.0f)" % (k, v, flops / v / 1e9, 3 * flops / v / 1e9) for k, v in res.items()))
