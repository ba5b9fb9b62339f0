#!/usr/bin/env python
"""A/B of the weight-ring depth of conv3x3_patch_kernel (2 stages vs 3: a weight stage gets two stage times to land) on
single-level launches narrow enough for the 3-stage ring to fit (image width <= 125), uniform 256-position tiles, interleaved
rounds; outputs must be bit-identical.  Also: where do the GN statistics of a mixed launch differ from the uniform one."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H, _lib
RING2, UNIFORM = 0x08000000, 0x4000
dev = torch.device("cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, B, sizes in (("P4 50x84 B=16", 16, [(50, 84)]), ("P4 50x84 B=30", 30, [(50, 84)]), ("100x120 B=4", 4, [(100, 120)]),
                       ("P4-P7 B=16", 16, [(50, 84), (25, 42), (13, 21), (7, 11)])):
    lv = H.Levels(B, sizes)
    x = (torch.randn(lv.rows, 256, device=dev) * 0.5).to(torch.bfloat16)
    w = torch.randn(256, 256, 3, 3, device=dev) / 48
    wp, cpp = H.prep_conv_weight_patch(w)
    ys = {k: torch.zeros(lv.rows, 256, dtype=torch.bfloat16, device=dev) for k in ("ring3", "ring2")}
    ds = {"ring3": H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, 256, cpp, 3, 1, 1, 256, 256, flags=UNIFORM),
          "ring2": H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, 256, cpp, 3, 1, 1, 256, 256, flags=UNIFORM | RING2)}
    res = {k: [] for k in ds}
    for rnd in range(6):
        for k in ds:
            e0.record()
            for _ in range(10):
                H.conv3x3_patch(ds[k], x, wp, None, ys[k])
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[k].append(e0.elapsed_time(e1) / 10)
    flops = 2.0 * lv.rows * 256 * 2304
    tiles = H.conv3x3_patch_plan(ds["ring3"])
    print("%-14s %7.2f GFLOP " % (name, flops / 1e9) + "  ".join(
        "%s %.4f ms %.0f TF/s" % (k, sorted(v)[len(v) // 2], flops / sorted(v)[len(v) // 2] / 1e9) for k, v in res.items()),
        " bit-identical:", bool(torch.equal(ys["ring3"], ys["ring2"])), tiles, flush=True)

# ---- GN statistics: mixed vs uniform launch, which entries differ (none since the sum of squares is an explicit fma)
sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
B, C = 2, 256
lv = H.Levels(B, sizes)
g = torch.Generator().manual_seed(15)
x = (torch.randn(lv.rows, C, generator=g) * 0.5).to(torch.bfloat16).to(dev)
w = (torch.randn(C, C, 3, 3, generator=g) / 48).to(dev)
wp, cpp = H.prep_conv_weight_patch(w)
S = 2 * B * len(sizes) * (C // 8)
st = {}
for flag in (0, UNIFORM):
    y = torch.zeros(lv.rows, C, dtype=torch.bfloat16, device=dev)
    s = torch.zeros(S, dtype=torch.int64, device=dev)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, C, C, cpp, 3, 1, 1, C, C, flags=flag)
    H.conv3x3_patch(d, x, wp, None, y, s)
    torch.cuda.synchronize()
    st[flag] = s.view(B, len(sizes), C // 8, 2).cpu()
    print("plan", flag, H.conv3x3_patch_plan(d))
diff = (st[0] - st[UNIFORM])
idx = diff.nonzero()
print("GN entries that differ:", idx.shape[0], "of", S)
for i in idx[:24].tolist():
    print(i, int(diff[tuple(i)]), int(st[0][tuple(i)]))
