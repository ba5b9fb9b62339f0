#!/usr/bin/env python
"""EXPERIMENT (needs `make -C sipmask_amd/csrc EXPERIMENTS=1`): the 256 x 256 patch-conv tile on 4 waves (128 x 128 per wave,
accumulators in AGPRs, one wave per SIMD) against the shipped 8-wave tile, uniform launches, compiler-scheduled and
software-pipelined stage; interleaved rounds, bit equality of the outputs checked."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
UNIFORM, PIPE, W4 = 0x4000, 0x800, 0x10000000
dev = torch.device("cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, B, sizes, groups in (("tower pair B=2 (the benchmarked launch)", 2, LEVELS, 2), ("tower pair B=4", 4, LEVELS, 2),
                               ("one full round: 100x120 B=5", 5, [(100, 120)], 1), ("fpn.out0 B=2", 2, LEVELS[:1], 1)):
    lv = H.Levels(B, sizes)
    x = (torch.randn(lv.rows, 256, device=dev) * 0.5).to(torch.bfloat16)
    ws = [torch.randn(256, 256, 3, 3, device=dev) / 48 for _ in range(groups)]
    packed = [H.prep_conv_weight_patch(w)[0] for w in ws]
    wq = torch.stack(packed).contiguous()
    variants = {"w8": UNIFORM, "w8_pipe": UNIFORM | PIPE, "w4": UNIFORM | W4, "w4_pipe": UNIFORM | W4 | PIPE}
    ys = {k: torch.zeros(groups * lv.rows, 256, dtype=torch.bfloat16, device=dev) for k in variants}
    ds = {k: H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, 256, 256, 3, 1, 1, 256, 256, flags=f, ngroups=groups,
                              x_group_rows=0, y_group_rows=lv.rows, w_group_stride=packed[0].numel(), bias_group_stride=0,
                              gn_group_stride=0) for k, f in variants.items()}
    res = {k: [] for k in variants}
    for rnd in range(6):
        for k in variants:
            e0.record()
            for _ in range(10):
                H.conv3x3_patch(ds[k], x, wq, None, ys[k])
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[k].append(e0.elapsed_time(e1) / 10)
    flops = 2.0 * lv.rows * 256 * 2304 * groups
    print("%-40s %7.2f GFLOP " % (name, flops / 1e9) + "  ".join(
        "%s %.4f ms %.0f TF/s" % (k, sorted(v)[len(v) // 2], flops / sorted(v)[len(v) // 2] / 1e9) for k, v in res.items()),
        " bit-identical to w8:", {k: bool(torch.equal(ys[k], ys["w8"])) for k in variants if k != "w8"}, flush=True)
