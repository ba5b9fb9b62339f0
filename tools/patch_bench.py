#!/usr/bin/env python
# NOTE (round 3): the A/B flags this tool toggles are experiments -- build the library with `make -C sipmask_amd/csrc EXPERIMENTS=1` first (csrc/experiments.h); the default build ignores them.
"""patch-resident 3x3 kernel vs the implicit-GEMM kernel on the R50 head shapes (B=2 and B=4), interleaved rounds"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H, _lib
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
dev = torch.device("cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in (1, 2, 4):
    for name, sizes, co, f32 in (("tower x5lev", LEVELS, 256, False), ("fpn.out0", LEVELS[:1], 256, False), ("cls_cof x5lev", LEVELS, 208, True)):
        lv = H.Levels(B, sizes)
        x = (torch.randn(lv.rows, 256, device=dev) * 0.5).to(torch.bfloat16)
        w = torch.randn(co, 256, 3, 3, device=dev) / 48
        wq, cp = H.prep_conv_weight(w, 256)
        wp, cpp = H.prep_conv_weight_patch(w)
        y = torch.empty(lv.rows, co, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
        fl = _lib.SM_CONV_OUT_F32 if f32 else 0
        d1 = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, co, cp, 3, 1, 1, 256, co, flags=fl)
        d2 = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, co, cp, 3, 1, 1, 256, co, flags=fl | 0x00440000)
        d3 = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, co, cpp, 3, 1, 1, 256, co, flags=fl)
        d4 = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, co, cpp, 3, 1, 1, 256, co, flags=fl | 0x4000)
        d5 = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, co, cpp, 3, 1, 1, 256, co, flags=fl | 0x2000)
        d6 = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, co, cpp, 3, 1, 1, 256, co, flags=fl | 0x1000)
        d7 = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, co, cpp, 3, 1, 1, 256, co, flags=fl | 0x800)
        d8 = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, co, cpp, 3, 1, 1, 256, co, flags=fl | 0x4800)
        fns = {"igemm128": lambda: H.conv2d(d1, x, wq, None, None, y), "igemm256": lambda: H.conv2d(d2, x, wq, None, None, y),
               "patch": lambda: H.conv3x3_patch(d3, x, wp, None, y), "patch_uniform": lambda: H.conv3x3_patch(d4, x, wp, None, y),
               "patch_128": lambda: H.conv3x3_patch(d5, x, wp, None, y), "patch_192": lambda: H.conv3x3_patch(d6, x, wp, None, y),
               "pipe": lambda: H.conv3x3_patch(d7, x, wp, None, y), "pipe_uniform": lambda: H.conv3x3_patch(d8, x, wp, None, y)}
        res = {k: [] for k in fns}
        for rnd in range(6):
            for k, fn in fns.items():
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    res[k].append(e0.elapsed_time(e1) / 10)
        flops = 2.0 * lv.rows * co * 2304
        print("B=%d %-14s %7.2f GFLOP  " % (B, name, flops / 1e9) + "   ".join(
            "%s %.4f ms %.0f TF/s" % (k, sorted(v)[len(v) // 2], flops / sorted(v)[len(v) // 2] / 1e9) for k, v in res.items()),
            " plan", H.conv3x3_patch_plan(d3), "128:", H.conv3x3_patch_plan(d5), "192:", H.conv3x3_patch_plan(d6))
