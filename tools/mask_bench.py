#!/usr/bin/env python
"""sm_mask_assemble_lo alone, at BASELINE's geometry (4 images, 100 detections each, basis 100 x 168 x 32, masks 800 x 1344):
the worst case (every box the whole image), COCO-sized boxes (bench.coco_boxes) and the alternation of the two box sets a
pipelined slot sees (old rectangle zeroed, new one written).  HIP events over `--iters` launches.

    python tools/mask_bench.py [--iters 20] [--only worst]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sipmask_amd import hip_ops as H  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--lib", type=str, default="", help="another build of libsipmask_hip.so (A/B of the mask kernel)")
    args = ap.parse_args()
    if args.lib:
        from sipmask_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    dev = torch.device("cuda")
    B, n, kmax = 4, 100, 1000
    IH, IW, h0, w0 = 800, 1344, 100, 168
    g = torch.Generator().manual_seed(3)
    basis = torch.randn(B, h0, w0, 32, generator=g).to(dev)
    cofs = (torch.randn(B, kmax, 128, generator=g) * 0.5).to(dev)
    keep = torch.arange(n, dtype=torch.int64, device=dev).repeat(B, 1).contiguous()
    ndet = torch.full((B,), n, dtype=torch.int32, device=dev)
    sets, _ = bench.coco_boxes(3, B, n, IH, IW)
    cases = {}
    whole = torch.zeros(B, n, 5)
    whole[..., 2], whole[..., 3], whole[..., 4] = float(IW), float(IH), 0.9
    cases["worst"] = [whole]
    cases["coco"] = [torch.cat([sets[0], torch.full((B, n, 1), 0.9)], -1)]
    cases["coco_alternating"] = [torch.cat([s, torch.full((B, n, 1), 0.9)], -1) for s in sets]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, dets in cases.items():
        if args.only and args.only not in name:
            continue
        dets = [d.to(dev).contiguous() for d in dets]
        buf = H.mask_assemble_lo_alloc(B, n, IH, IW, dev)
        run = lambda i: H.mask_assemble_lo(basis, h0, w0, 4, cofs, keep, dets[i % len(dets)], ndet, IH, IW, 1.0, 2.0, 2.0, 0.4, buf)
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        e0.record()
        for i in range(args.iters):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        px = sum(float(((d[..., 2] - d[..., 0]) * (d[..., 3] - d[..., 1])).sum()) for d in dets) / len(dets)
        print("%-18s %8.4f ms per launch   %6.1f M box pixels   masks set %.4f   sum %d" %
              (name, ms, px / 1e6, float((buf["masks"] != 0).float().mean()), int(buf["masks"].sum(dtype=torch.int64))))
        del buf


if __name__ == "__main__":
    main()
