#!/usr/bin/env python
"""the tower weight gradient (3x3 256 -> 256 over the B=4 pyramid) through sm_wgrad_direct, 20 launches: target of PMC passes"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
dev = torch.device("cuda")
B, ci, co = 4, 256, 256
lv = H.Levels(B, LEVELS)
x = (torch.randn(lv.rows, ci, device=dev) * 0.5).to(torch.bfloat16)
g = (torch.randn(lv.rows, co, device=dev) * 0.1).to(torch.bfloat16)
gw = torch.empty(9 * ci, co, device=dev)
d = H.make_conv_desc(B, LEVELS, LEVELS, lv.row0, lv.row0, ci, co, co, 3, 1, 1, ci, co, flags=256)
for _ in range(20):
    H.conv2d_bwd(d, x, None, None, g, None, gw, None)
torch.cuda.synchronize()
print("ok")
