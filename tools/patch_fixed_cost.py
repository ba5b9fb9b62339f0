#!/usr/bin/env python
"""How much of a patch-kernel tile is NOT its K loop: the grouped tower launch (B = 4, five FPN levels, 2 x 256 couts) timed with
64 / 128 / 256 / 512 input channels (1 / 2 / 4 / 8 channel-chunk pairs of nine weight stages each).  A linear fit of the launch time
over the pair count gives the per-pair time and the fixed part per launch (dispatch, first patch chunk + weight stage, epilogue with
the GroupNorm statistics, tile-count rounding).

    python tools/patch_fixed_cost.py [--iters 30]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H, _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda")
    B, G, CO = 4, 2, 256
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    lv = H.Levels(B, sizes)
    res = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for cin in (64, 128, 256, 512, 256, 128, 64):
        x = (torch.randn(lv.rows, cin, device=dev) * 0.5).to(torch.bfloat16)
        ws = [torch.randn(CO, cin, 3, 3, device=dev) / (9 * cin) ** 0.5 for _ in range(G)]
        wq = torch.stack([H.prep_conv_weight_patch(w)[0] for w in ws]).contiguous()
        bias = torch.randn(G, CO, device=dev)
        S = 2 * B * len(sizes) * (CO // 8)
        d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, cin, CO, 256, 3, 1, 1, cin, CO, ngroups=G, x_group_rows=0,
                             y_group_rows=lv.rows, w_group_stride=wq[0].numel(), bias_group_stride=CO, gn_group_stride=S)
        y = torch.empty(G * lv.rows, CO, dtype=torch.bfloat16, device=dev)
        stats = torch.zeros(G * S, dtype=torch.int64, device=dev)
        assert H.conv3x3_patch_supported(d)
        pl = H.conv3x3_patch_plan(d)
        for _ in range(3):
            H.conv3x3_patch(d, x, wq, bias, y, stats)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.iters):
            H.conv3x3_patch(d, x, wq, bias, y, stats)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        fl = 2.0 * G * lv.rows * CO * cin * 9
        res.append((cin, ms))
        print("cin %4d  pairs %d  %.4f ms  %7.1f TFLOP/s   blocks %s" % (cin, cin // 64, ms, fl / ms / 1e9, pl.get("blocks", pl)))
    # the epilogue's share: the cin = 256 launch without the fused GroupNorm statistics (no butterflies, LDS / global atomics,
    # barriers, and no zero-fill launch in front), interleaved with the full launch
    cin = 256
    x = (torch.randn(lv.rows, cin, device=dev) * 0.5).to(torch.bfloat16)
    ws = [torch.randn(CO, cin, 3, 3, device=dev) / (9 * cin) ** 0.5 for _ in range(G)]
    wq = torch.stack([H.prep_conv_weight_patch(w)[0] for w in ws]).contiguous()
    bias = torch.randn(G, CO, device=dev)
    S = 2 * B * len(sizes) * (CO // 8)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, cin, CO, 256, 3, 1, 1, cin, CO, ngroups=G, x_group_rows=0,
                         y_group_rows=lv.rows, w_group_stride=wq[0].numel(), bias_group_stride=CO, gn_group_stride=S)
    y = torch.empty(G * lv.rows, CO, dtype=torch.bfloat16, device=dev)
    stats = torch.zeros(G * S, dtype=torch.int64, device=dev)
    for rep in range(3):
        for name, st in (("with statistics", stats), ("without", None)):
            for _ in range(3):
                H.conv3x3_patch(d, x, wq, bias, y, st)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                H.conv3x3_patch(d, x, wq, bias, y, st)
            e1.record()
            torch.cuda.synchronize()
            print("cin 256 %-16s %.4f ms" % (name, e0.elapsed_time(e1) / args.iters))
    xs = np.array([c // 64 for c, _ in res], float)
    ys = np.array([m for _, m in res])
    k, b = np.polyfit(xs, ys, 1)
    print("fit: %.4f ms per pair of channel chunks + %.4f ms fixed per launch; at cin 256 the fixed part is %.1f %% of the launch"
          % (k, b, 100 * b / (4 * k + b)))


if __name__ == "__main__":
    main()
