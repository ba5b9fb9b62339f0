#!/usr/bin/env python
"""how much of the training step is host time: python returns from (forward + loss + backward + SGD enqueue) after T_host,
the device finishes after T_total"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import bench
from sipmask_amd.synthetic import build_synthetic_detector
from sipmask_amd.dist_train import HipSGD
from sipmask_amd import ops_rows
dev = torch.device("cuda")
det = build_synthetic_detector(50, seed=0).to(dev); det.train()
B, Hh, Ww = 4, 800, 1344
img = torch.randn(B, 3, Hh, Ww).to(dev)
gtb, gtl, gtm = bench.synthetic_gt(0, B, Hh, Ww, dev)
metas = [dict(img_shape=(Hh, Ww, 3), pad_shape=(Hh, Ww, 3), scale_factor=1.0) for _ in range(B)]
opt = HipSGD(det.named_parameters(), lr=0.0005, momentum=0.9, weight_decay=1e-4)
def step(parts):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(); ops_rows.begin_step()
    losses = det.forward_train(img, metas, gtb, gtl, gt_masks=gtm)
    t1 = time.perf_counter()
    sum(losses.values()).backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    parts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
parts = []
for _ in range(8): step(parts)
p = np.array(parts[3:]).mean(0) * 1e3
print("host: forward+loss %.1f ms, backward %.1f ms, sgd %.1f ms; device tail after python returned %.1f ms; total %.1f ms" % tuple(p))
