#!/usr/bin/env python
"""EXPERIMENT (needs `make -C sipmask_amd/csrc EXPERIMENTS=1`): how the time of ONE round of 256-position patch-conv tiles
depends on the number of busy CUs, for the full kernel and for its two ablations (no LDS-DMA in the main loop / no MFMA and
fragment reads: wrong results by construction).  If the MFMA-only tile slows down with more CUs the shared limiter is the
clock; if only the DMA-carrying variants do, it is the L2 -> LDS fabric."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H
UNIFORM, NO_DMA, NO_MFMA = 0x4000, 0x200, 0x100
dev = torch.device("cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
w = torch.randn(256, 256, 3, 3, device=dev) / 48
wp, cpp = H.prep_conv_weight_patch(w)
print("# tiles = one 256x256 tile per CU; times in us per launch (median of 5 x 20 launches)")
for tiles_target in (32, 64, 128, 192, 224, 256):
    # images of 30 x 126 (128 padded columns): 30 * 128 / 256 = 15 tiles per image
    B = max(1, tiles_target // 15)
    sizes = [(30, 126)]
    lv = H.Levels(B, sizes)
    x = (torch.randn(lv.rows, 256, device=dev) * 0.5).to(torch.bfloat16)
    y = torch.zeros(lv.rows, 256, dtype=torch.bfloat16, device=dev)
    row = []
    for name, fl in (("full", UNIFORM), ("no_dma", UNIFORM | NO_DMA), ("no_mfma", UNIFORM | NO_MFMA)):
        d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, 256, cpp, 3, 1, 1, 256, 256, flags=fl)
        ntile = H.conv3x3_patch_plan(d)["big"]
        ts = []
        for rnd in range(6):
            e0.record()
            for _ in range(20):
                H.conv3x3_patch(d, x, wp, None, y)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        row.append("%s %.1f" % (name, sorted(ts)[2]))
    print("tiles %3d (B=%2d)  " % (ntile, B) + "   ".join(row), flush=True)
