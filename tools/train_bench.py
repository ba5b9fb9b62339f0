#!/usr/bin/env python
"""First measurement of BASELINE config #4 on this code base: SipMask-R50 training step (forward_train + loss +
backward + bucketed all-reduce + SGD), 4 images of 3x800x1344 per GPU, synthetic ground truth.

    python tools/train_bench.py [--steps 3] [--batch 4]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py   (DDP over RCCL)

The training graph runs layer by layer on the HIP autograd ops (NCHW<->NHWC conversion in every conv, f32 saved
activations): it is the correct-first version, not the fused plan the inference path has.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1344)
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    from sipmask_amd.synthetic import build_synthetic_detector
    from sipmask_amd.dist_train import GradBucketer, HipSGD, detector_train_step
    det = build_synthetic_detector(50, seed=0).to(dev)
    det.train()
    B, Hh, Ww = args.batch, args.height, args.width
    g = torch.Generator().manual_seed(100 + rank)
    img = torch.randn(B, 3, Hh, Ww, generator=g).to(dev)
    rng = np.random.RandomState(rank)
    gtb, gtl, gtm = [], [], []
    yy, xx = np.mgrid[:Hh, :Ww]
    for _ in range(B):
        n = 6
        xy = rng.rand(n, 2) * np.array([Ww * 0.6, Hh * 0.6])
        wh = rng.rand(n, 2) * np.array([Ww * 0.35, Hh * 0.35]) + 24
        b = np.concatenate([xy, np.minimum(xy + wh, [Ww - 1, Hh - 1])], 1).astype(np.float32)
        m = np.zeros((n, Hh, Ww), np.uint8)
        for k in range(n):
            cx, cy, rx, ry = (b[k, 0] + b[k, 2]) / 2, (b[k, 1] + b[k, 3]) / 2, (b[k, 2] - b[k, 0]) / 2, (b[k, 3] - b[k, 1]) / 2
            m[k] = (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0
        gtb.append(torch.from_numpy(b).to(dev))
        gtl.append(torch.from_numpy(rng.randint(1, 81, n).astype(np.int64)).to(dev))
        gtm.append(m)
    metas = [dict(img_shape=(Hh, Ww, 3), pad_shape=(Hh, Ww, 3), scale_factor=1.0) for _ in range(B)]
    opt = HipSGD(det.named_parameters(), lr=0.0005, momentum=0.9, weight_decay=1e-4)
    bucket = GradBucketer([p for p in det.parameters() if p.requires_grad]) if world > 1 else None
    losses = detector_train_step(det, img, metas, gtb, gtl, gtm, opt, bucket)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = detector_train_step(det, img, metas, gtb, gtl, gtm, opt, bucket)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if rank == 0:
        print(json.dumps({"metric": "SipMask-R50 training step (fwd+bwd+allreduce+SGD), unfused HIP autograd ops",
                          "ms_per_step": round(dt * 1e3, 1), "img_per_s": round(B * world / dt, 2), "n_gpus": world,
                          "batch_per_gpu": B, "image": [3, Hh, Ww], "losses": {k: round(v, 4) for k, v in losses.items()},
                          "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
