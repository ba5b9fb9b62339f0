#!/usr/bin/env python
"""CPU emulation of the split-bf16 ("x3") head plan, to size its error BEFORE building kernels.

x3 arithmetic: every conv operand is split into two bf16 halves, v = hi + lo (hi = bf16(v), lo = bf16(v - hi): 16
mantissa bits together), and a product is three MFMA terms  hi*hi + hi*lo + lo*hi  accumulated in f32; activations
between the head's layers stay f32.  This script runs the ORACLE's head (oracle/model.py: head_forward) twice on the same
fp32 FPN features of one 800x1344 image -- once as is, once with every F.conv2d / deform_conv operand replaced by that
three-term product -- and reports the error of the coefficient / basis tensors and of the mask logits at the top-100
positions.  Test infrastructure only (imports oracle/)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model as OM   # noqa: E402
from oracle import ops as O      # noqa: E402


HALF = torch.bfloat16          # operand half type: bf16 (8+8 mantissa bits) or f16 (11+11; weights pre-scaled, see wscale)


def split(t):
    hi = t.to(HALF).float()
    lo = (t - hi).to(HALF).float()
    return hi, lo


def wscale(w):
    """f16 halves: the low half of a weight of ~1e-2 is ~5e-6, deep in f16's subnormals (quantum 6e-8) -- scale the
    weight matrix by a power of two so that its largest element sits near 2^12 (exact; undone on the accumulator)"""
    if HALF != torch.float16:
        return 1.0
    import math
    m = float(w.abs().max())
    return 2.0 ** math.floor(math.log2(4096.0 / m)) if m > 0 else 1.0


def main(terms=3, hw=(800, 1344)):
    from sipmask_amd.synthetic import build_synthetic_detector
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    det = build_synthetic_detector(50, seed=0)
    sd = {k: v.detach().float().cpu().clone() for k, v in det.state_dict().items()}
    img = torch.randn(1, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        pyr = OM.fpn_forward(sd, OM.backbone_forward(sd, img, 50))
        t0 = time.time()
        ref = OM.head_forward(sd, pyr)
        print("f32 head: %.1f s" % (time.time() - t0))
        real_conv, real_deform = F.conv2d, O.deform_conv

        def conv_x3(x, w, b=None, stride=1, padding=0, *a, **k):
            xh, xl = split(x)
            sw = wscale(w)
            wh, wl = split(w * sw)
            y = real_conv(xh, wh, None, stride, padding, *a, **k)
            y = y + real_conv(xh, wl, None, stride, padding, *a, **k) + real_conv(xl, wh, None, stride, padding, *a, **k)
            if terms == 4:
                y = y + real_conv(xl, wl, None, stride, padding, *a, **k)
            y = y / sw
            return y if b is None else y + b.view(1, -1, 1, 1)

        def deform_x3(x, offset, w, stride, pad, dil, groups, **k):
            # the sampled column is blended in f32 from the f32 activation and THEN split (csrc: the blend's result is the
            # MFMA operand); linear in w and in the column, so three calls with rounded halves reproduce it
            sw = wscale(w)
            wh, wl = split(w * sw)
            hi = lambda t: t.to(HALF).float()
            lo = lambda t: (t - t.to(HALF).float()).to(HALF).float()
            return (real_deform(x, offset, wh, stride, pad, dil, groups, col_round=hi) +
                    real_deform(x, offset, wl, stride, pad, dil, groups, col_round=hi) +
                    real_deform(x, offset, wh, stride, pad, dil, groups, col_round=lo)) / sw
        OM.F.conv2d, OM.ops.deform_conv = conv_x3, deform_x3
        try:
            t0 = time.time()
            got = OM.head_forward(sd, pyr)
            print("x3 head: %.1f s" % (time.time() - t0))
        finally:
            OM.F.conv2d, OM.ops.deform_conv = real_conv, real_deform
    names = ("cls", "bbox", "ctr", "cof")
    for n, a, b in zip(names, got[:4], ref[:4]):
        e = max(float((x - y).abs().max()) for x, y in zip(a, b))
        m = max(float(y.abs().max()) for y in b)
        print("%-5s max_abs %.3e  (ref max %.3g)" % (n, e, m))
    print("basis max_abs %.3e (ref max %.3g)" % (float((got[4] - ref[4]).abs().max()), float(ref[4].abs().max())))
    # mask logits at the 100 best-scoring positions (all four quadrants, every pixel of the basis)
    sc = torch.cat([(c[0].sigmoid().max(0)[0] * t[0, 0].sigmoid()).reshape(-1) for c, t in zip(ref[0], ref[2])])
    top = sc.topk(100)[1]
    cof_r = torch.cat([c[0].reshape(128, -1) for c in ref[3]], 1)[:, top]
    cof_g = torch.cat([c[0].reshape(128, -1) for c in got[3]], 1)[:, top]
    br, bg = ref[4][0].reshape(32, -1).t(), got[4][0].reshape(32, -1).t()
    worst, big = 0.0, 0.0
    for q in range(4):
        lr = br @ cof_r[32 * q:32 * q + 32]
        lg = bg @ cof_g[32 * q:32 * q + 32]
        worst = max(worst, float((lr - lg).abs().max()))
        big = max(big, float(lr.abs().max()))
    print("mask logits (top-100 positions): max_abs %.3e (ref max %.3g)  terms=%d" % (worst, big, terms))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "f16":
        HALF = torch.float16
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
