#!/usr/bin/env python
"""weight gradient of the training step's conv shapes (B=4, 800x1344): sm_wgrad_direct vs the im2col^T GEMM path"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H
dev = torch.device("cuda")
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
SHAPES = [("tower 3x3 256->256 x5lev", 256, 256, 3, 1, 1, LEVELS),
          ("fpn.out0 3x3 256->256", 256, 256, 3, 1, 1, LEVELS[:1]),
          ("layer2 3x3 128->128", 128, 128, 3, 1, 1, [(100, 168)]),
          ("layer2 1x1 512->128", 512, 128, 1, 1, 0, [(100, 168)]),
          ("layer2 1x1 128->512", 128, 512, 1, 1, 0, [(100, 168)]),
          ("layer3 3x3 256->256", 256, 256, 3, 1, 1, [(50, 84)]),
          ("layer3 1x1 1024->256", 1024, 256, 1, 1, 0, [(50, 84)]),
          ("layer4 3x3 512->512", 512, 512, 3, 1, 1, [(25, 42)]),
          ("layer4 1x1 512->2048", 512, 2048, 1, 1, 0, [(25, 42)]),
          ("mask lat0 1x1 768->512", 768, 512, 1, 1, 0, [(100, 168)])]
B = 4
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = {"direct": 0.0, "direct128": 0.0, "gemm": 0.0}
for name, ci, co, k, s, p, sizes in SHAPES:
    lv = H.Levels(B, sizes)
    x = (torch.randn(lv.rows, ci, device=dev) * 0.5).to(torch.bfloat16)
    g = (torch.randn(lv.rows, co, device=dev) * 0.1).to(torch.bfloat16)
    gw = torch.empty(k * k * ci, co, device=dev)
    res = {}
    for label, flag in (("direct", 256), ("direct128", 256 | 512), ("gemm", 128)):
        d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, ci, co, co, k, s, p, ci, co, flags=flag)
        ts = []
        for r in range(5):
            e0.record()
            for _ in range(3):
                H.conv2d_bwd(d, x, None, None, g, None, gw, None)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) / 3)
        res[label] = sorted(ts)[len(ts) // 2]
        tot[label] += res[label]
    fl = 2.0 * lv.rows * co * ci * k * k
    print("%-28s %7.1f GFLOP  direct %.4f ms (%4.0f TF/s)   direct, 128x128 tile only %.4f ms (%4.0f TF/s)   gemm path %.4f ms (%4.0f TF/s)" % (
        name, fl / 1e9, res["direct"], fl / res["direct"] / 1e9, res["direct128"], fl / res["direct128"] / 1e9, res["gemm"], fl / res["gemm"] / 1e9))
print("sum: direct %.3f ms, direct 128x128 only %.3f ms, gemm path %.3f ms" % (tot["direct"], tot["direct128"], tot["gemm"]))
