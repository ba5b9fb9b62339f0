#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM conv kernel on the shapes that dominate SipMask-R50
(batch 4, 800x1344).  Interleaved rounds in one process (guide rule 24); prints TFLOP/s and GB/s.

    python tools/conv_bench.py [--rounds 5] [--flags 0]
"""
import argparse
import sys, os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H, _lib  # noqa: E402

B = 4
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
# name, sizes, cin, cout, k, stride, pad, flags, residual
SHAPES = [
    ("tower 3x3 256->256 x5lev", LEVELS, 256, 256, 3, 1, 1, 0, False),
    ("fpn.out0 3x3 256->256 100x168", LEVELS[:1], 256, 256, 3, 1, 1, 0, False),
    ("cls_cof 3x3 256->208 f32 x5lev", LEVELS, 256, 208, 3, 1, 1, _lib.SM_CONV_OUT_F32, False),
    ("l1.conv3 1x1 64->256 +res 200x336", [(200, 336)], 64, 256, 1, 1, 0, _lib.SM_CONV_RELU, True),
    ("l1.conv2 3x3 64->64 200x336", [(200, 336)], 64, 64, 3, 1, 1, _lib.SM_CONV_RELU, False),
    ("l1.conv1 1x1 256->64 200x336", [(200, 336)], 256, 64, 1, 1, 0, _lib.SM_CONV_RELU, False),
    ("l2.conv2 3x3 128->128 100x168", [(100, 168)], 128, 128, 3, 1, 1, _lib.SM_CONV_RELU, False),
    ("l2.conv3 1x1 128->512 +res 100x168", [(100, 168)], 128, 512, 1, 1, 0, _lib.SM_CONV_RELU, True),
    ("l3.conv2 3x3 256->256 50x84", [(50, 84)], 256, 256, 3, 1, 1, _lib.SM_CONV_RELU, False),
    ("l3.conv3 1x1 256->1024 +res 50x84", [(50, 84)], 256, 1024, 1, 1, 0, _lib.SM_CONV_RELU, True),
    ("l3.conv1 1x1 1024->256 50x84", [(50, 84)], 1024, 256, 1, 1, 0, _lib.SM_CONV_RELU, False),
    ("l2.conv1 1x1 512->128 100x168", [(100, 168)], 512, 128, 1, 1, 0, _lib.SM_CONV_RELU, False),
    ("l4.conv3 1x1 512->2048 +res 25x42", [(25, 42)], 512, 2048, 1, 1, 0, _lib.SM_CONV_RELU, True),
    ("l4.conv1 1x1 2048->512 25x42", [(25, 42)], 2048, 512, 1, 1, 0, _lib.SM_CONV_RELU, False),
    ("fpn.lat2 1x1 2048->256 25x42", [(25, 42)], 2048, 256, 1, 1, 0, 0, False),
    ("l3.ds 1x1 s2 512->1024 100x168", [(100, 168)], 512, 1024, 1, 2, 0, 0, False),
    ("l4.ds 1x1 s2 1024->2048 50x84", [(50, 84)], 1024, 2048, 1, 2, 0, 0, False),
    ("l4.conv2 3x3 512->512 25x42", [(25, 42)], 512, 512, 3, 1, 1, _lib.SM_CONV_RELU, False),
    ("mask_lat0 1x1 768->512 100x168", [(100, 168)], 768, 512, 1, 1, 0, _lib.SM_CONV_RELU, False),
    ("stem 7x7 s2 8->64 800x1344", [(800, 1344)], 8, 64, 7, 2, 3, _lib.SM_CONV_RELU, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", type=str, default="", help="comma separated substring filters on the shape name")
    ap.add_argument("--variants", type=str, default="0,%d" % 0x40000000,
                    help="comma separated extra flag words; 'FLAGS:rN' also sets SIPMASK_EXP_K32_RING=N for that variant's "
                         "launches (experiments build)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    variants = args.variants.split(",")

    def vflags(v):
        return int(v.split(":")[0], 0)

    def vring(v):
        return int(v.split(":r")[1]) if ":r" in v else 0

    cases = []
    for name, sizes, cin, cout, k, s, p, flags, res in SHAPES:
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        lv = H.Levels(B, sizes)
        osz = [((h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1) for h, w in sizes]
        lo = H.Levels(B, osz)
        x = (torch.randn(lv.rows, cin, device=dev) * 0.5).to(torch.bfloat16)
        w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        wq, co_pad = H.prep_conv_weight(w, cin)
        f32 = bool(flags & _lib.SM_CONV_OUT_F32)
        y = torch.empty(lo.rows, cout, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
        r = (torch.randn(lo.rows, cout, device=dev)).to(torch.bfloat16) if res else None
        bias = torch.randn(cout, device=dev)
        descs = []
        for v in variants:
            fl = flags | vflags(v) | (_lib.SM_CONV_RES_ADD if res else 0)
            descs.append(H.make_conv_desc(B, sizes, osz, lv.row0, lo.row0, cin, cout, co_pad, k, s, p, cin, cout,
                                          flags=fl, res_cstride=cout))
        flops = 2.0 * lo.rows * cout * cin * k * k
        byts = lv.rows * cin * 2 + lo.rows * cout * (4 if f32 else 2) * (2 if res else 1) + wq.numel() * 2
        # every variant must reproduce variant 0 bit for bit (same K order per output)
        plans, y0 = [], None
        for v, d in zip(variants, descs):
            os.environ["SIPMASK_EXP_K32_RING"] = str(vring(v))
            pl = H.conv_plan(d)
            plans.append("%dx%d k%d%s b%d" % (pl["tile_cout"], pl["tile_pos"], pl["k_step"],
                                            (" ring%d" % pl["ring_stages"]) if pl["ring_stages"] else "", pl["blocks"]))
            y.zero_()
            H.conv2d(d, x, wq, bias, r, y)
            torch.cuda.synchronize()
            if y0 is None:
                y0 = y.clone()
            elif not torch.equal(y0, y):
                print("MISMATCH %s variant %s: max abs %.4g" % (name, v, (y0.float() - y.float()).abs().max().item()))
        print("%-40s plans: %s" % (name, " | ".join(plans)))
        cases.append((name, descs, x, wq, bias, r, y, flops, byts))
    res_ms = {(c[0], v): [] for c in cases for v in variants}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rnd in range(args.rounds + 1):
        for name, descs, x, wq, bias, r, y, flops, byts in cases:
            for v, d in zip(variants, descs):
                os.environ["SIPMASK_EXP_K32_RING"] = str(vring(v))
                e0.record()
                for _ in range(args.iters):
                    H.conv2d(d, x, wq, bias, r, y)
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    res_ms[(name, v)].append(e0.elapsed_time(e1) / args.iters)
    print("%-40s %10s %s" % ("shape", "GFLOP", "  ".join("%s: ms(med) TF/s GB/s" % v for v in variants)))
    for name, descs, x, wq, bias, r, y, flops, byts in cases:
        cols = []
        for v in variants:
            t = sorted(res_ms[(name, v)])
            med = t[len(t) // 2]
            cols.append("%8.4f %7.1f %6.0f" % (med, flops / med / 1e9, byts / med / 1e6))
        print("%-40s %10.2f   %s" % (name, flops / 1e9, "   |   ".join(cols)))


if __name__ == "__main__":
    main()
