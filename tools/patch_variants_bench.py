#!/usr/bin/env python
# NOTE (round 3): the A/B flags this tool toggles are experiments -- build the library with `make -C sipmask_amd/csrc EXPERIMENTS=1` first (csrc/experiments.h); the default build ignores them.
"""patch-conv main-loop variants and ablations on the tower shape (uniform 256-position tiles): B=2 is 184 tiles = one
partly filled round (pure per-tile time), B=4 368 tiles = two rounds"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
dev = torch.device("cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
VARS = (("base", 0x4000), ("pipe", 0x4800), ("stagger", 0x4400), ("stagger+pipe", 0x4C00), ("pingpong", 0x4080), ("ABL pp no-dma", 0x4280), ("ABL pp no-mfma", 0x4180), ("ABL no-dma", 0x4200), ("ABL no-mfma", 0x4100))
for B in (2, 4):
    lv = H.Levels(B, LEVELS)
    x = (torch.randn(lv.rows, 256, device=dev) * 0.5).to(torch.bfloat16)
    w = torch.randn(256, 256, 3, 3, device=dev) / 48
    wp, cpp = H.prep_conv_weight_patch(w)
    y = torch.empty(lv.rows, 256, dtype=torch.bfloat16, device=dev)
    descs = {n: H.make_conv_desc(B, LEVELS, LEVELS, lv.row0, lv.row0, 256, 256, cpp, 3, 1, 1, 256, 256, flags=f) for n, f in VARS}
    res = {n: [] for n, _ in VARS}
    for rnd in range(6):
        for n, _ in VARS:
            e0.record()
            for _ in range(10):
                H.conv3x3_patch(descs[n], x, wp, None, y)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[n].append(e0.elapsed_time(e1) / 10)
    flops = 2.0 * lv.rows * 256 * 2304
    print("B=%d tower (%d tiles): " % (B, H.conv3x3_patch_tiles(descs["base"])) + "   ".join(
        "%s %.4f ms (%.0f TF/s)" % (n, sorted(v)[len(v) // 2], flops / sorted(v)[len(v) // 2] / 1e9) for n, v in res.items()))
