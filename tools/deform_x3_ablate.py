#!/usr/bin/env python
"""Ablations of the x3 FeatureAlign window kernel (csrc/deform_patch_x3.hip; `make -C sipmask_amd/csrc clean && make EXPERIMENTS=1`):
what the K loop costs without the blend side, without the weight DMA, without the MFMAs.  B=4 head of the 800x1344 input."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H, _lib
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
dev = torch.device("cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
B = 4
lv = H.Levels(B, LEVELS)
x = (torch.randn(lv.rows, 256, device=dev) * 0.5).abs()
w = torch.randn(256, 256, 3, 3, device=dev) / 48
scale = H.x3_weight_scale([w])
w_win, cp = H.prep_deform_weight_x3(w, scale, 4)
y = torch.empty(lv.rows, 256, dtype=torch.float32, device=dev)
off = torch.randn(lv.rows, 72, device=dev) * float(sys.argv[1] if len(sys.argv) > 1 else 1.0)
NO_BLEND, NO_DMA, NO_MFMA = 0x400, 0x200, 0x100
for name, fl in (("full", 0), ("no blend side", NO_BLEND), ("no weight DMA", NO_DMA), ("no blend, no DMA", NO_BLEND | NO_DMA),
                 ("no MFMA / fragments", NO_MFMA), ("no MFMA, no DMA", NO_MFMA | NO_DMA)):
    d = H.make_conv_desc(B, LEVELS, LEVELS, lv.row0, lv.row0, 256, 256, cp, 3, 1, 1, 256, 256,
                         flags=_lib.SM_CONV_F16 | _lib.SM_CONV_OUT_F32 | fl, deform_groups=4, acc_scale=1.0 / scale)
    ts = []
    for rnd in range(4):
        e0.record()
        for _ in range(10):
            H.deform_conv2d_x3(d, x, off, w_win, None, y, None)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            ts.append(e0.elapsed_time(e1) / 10)
    print("%-22s %.4f ms" % (name, sorted(ts)[len(ts) // 2]))
