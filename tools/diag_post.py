#!/usr/bin/env python
"""diagnostic: candidate counts per (image, class) of the bench workload + GN/NMS timings"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd.synthetic import build_synthetic_detector, calibrate_cls_bias
B = 4
dev = torch.device("cuda")
det = build_synthetic_detector(50, seed=0)
img = torch.randn(B, 3, 800, 1344, generator=torch.Generator().manual_seed(1234)).to(dev)
eng = det.prepare(B, (800, 1344), (800, 1333, 3))
calibrate_cls_bias(det, eng, img, 1000)
eng = det.prepare(B, (800, 1344), (800, 1333, 3))
eng.run(img); torch.cuda.synchronize()
sc = eng.sel["scores"]          # [B][C][kmax]
n = (sc > 0.05).sum(-1).cpu()
print("per (image,class) count of scores > thr: max", int(n.max()), "mean %.1f" % float(n.float().mean()),
      "sorted top:", sorted(n.view(-1).tolist())[-12:], "zeros:", int((n == 0).sum()), "of", n.numel())
print("ncand", eng.sel["ncand"].cpu().tolist(), "ndet", eng.results()["ndet"].cpu().tolist())
