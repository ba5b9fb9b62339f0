#!/usr/bin/env python
"""FeatureAlign deformable conv backward at the BASELINE head shape (B=4, 5 levels): time of sm_deform_conv2d_bwd by requested output"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
dev = torch.device("cuda")
B, C, G = 4, 256, 4
lv = H.Levels(B, LEVELS)
g = torch.Generator().manual_seed(0)
x = (torch.randn(lv.rows, C, generator=g) * 0.5).to(torch.bfloat16).to(dev)
go = (torch.randn(lv.rows, C, generator=g) * 0.1).to(torch.bfloat16).to(dev)
w = (torch.randn(C, C, 3, 3, generator=g) / 48).to(dev)
for name, sc in (("offsets ~N(0,1.5)", 1.5), ("offsets = 0", 0.0), ("offsets ~N(0,8)", 8.0)):
    off = (torch.randn(lv.rows, G * 18, generator=g) * sc).to(dev)
    d = H.make_conv_desc(B, LEVELS, LEVELS, lv.row0, lv.row0, C, C, C, 3, 1, 1, C, C, deform_groups=G)
    w_t, _ = H.weight_prep(w, None, 2)
    gx = torch.empty(lv.rows, C, dtype=torch.float32, device=dev)
    goff = torch.empty_like(off)
    gw = torch.empty(9 * C, C, dtype=torch.float32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ds = H.make_conv_desc(B, LEVELS, LEVELS, lv.row0, lv.row0, C, C, C, 3, 1, 1, C, C, deform_groups=G, flags=L.SM_CONV_BWD_DX_SCATTER)
    for label, args, dd in (("gx+goff (gather + far scatter)", (gx, goff, None), d), ("gx+goff (atomic scatter alone)", (gx, goff, None), ds),
                            ("goff only", (None, goff, None), d), ("gw only", (None, None, gw), d), ("all", (gx, goff, gw), d)):
        ts = []
        for r in range(4):
            e0.record()
            H.deform_conv2d_bwd(dd, x, off, w_t, go, *args)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1))
        print("%-18s %-42s %.3f ms" % (name, label, sorted(ts)[1]))
