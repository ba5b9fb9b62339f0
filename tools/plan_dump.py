#!/usr/bin/env python
"""Prints the static launch plan of an engine configuration WITHOUT a GPU: every conv launch with the tile, K-step
width, K-loop variant and grid that `sm_conv_plan_query` (the launcher's own selection code) picks for it, the
fill of its last round of resident blocks, and the non-conv steps with their lanes.

    python tools/plan_dump.py [--batch 4] [--hw 800 1344] [--depth 50] [--variant r50|ssd|vis|benchmark|dcn] [--flags 0x...]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_on_cpu(variant="r50", batch=4, hw=(800, 1344), depth=50):
    """The plan is host logic: with torch.cuda.is_available patched the engine allocates its buffers on the CPU and
    prepares every descriptor; nothing is launched."""
    real = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    try:
        from oracle import model as OM
        from sipmask_amd.engine import SipMaskEngine
        kw = {}
        if variant == "ssd":
            sd = OM.init_state_dict(depth, 0, stacked_convs=2, norm=False)
            kw = dict(ssd_flag=True, scale_factor=[1.0, 1.0, 1.0, 1.0], rescale=True)
        elif variant == "dcn":
            sd = OM.init_state_dict(depth, 0, stacked_convs=2, norm=False, stage_with_dcn=(False, True, True, True), rescoring=True)
            kw = dict(ssd_flag=True, scale_factor=[1.0, 1.0, 1.0, 1.0], rescale=True)
        elif variant == "vis":
            from oracle import vis as OV
            sd = dict(OM.init_state_dict(depth, 0, num_classes=41, stacked_convs=3))
            sd.update(OV.init_vis_state_dict(0))
            kw = dict(vis=True, num_classes=41)
        elif variant == "benchmark":
            from oracle import fcos_core as OB
            from sipmask_amd.benchmark_variant import convert_state_dict
            sd = dict(OM.init_state_dict(depth, 0))
            sd = {k: v for k, v in sd.items() if not k.startswith("bbox_head.")}
            sd.update(convert_state_dict(OB.init_head_state_dict(0)))
            kw = dict(benchmark=dict(pre_nms_thresh=0.05, pre_nms_top_n=1000, nms_thresh=0.6, post_top_n=100))
        else:
            sd = OM.init_state_dict(depth, 0)
        return SipMaskEngine(sd, batch, tuple(hw), depth, device="cpu", **kw)
    finally:
        torch.cuda.is_available = real


def conv_rows(eng):
    from sipmask_amd import hip_ops as H
    rows = []
    for c in eng.convs:
        p = H.conv_plan(c.desc, deformable=c.offset is not None, with_gn_stats=c.gn_stats is not None)
        per_cu = 1 if p["threads"] == 512 else (4 if p["k_step"] == 32 else 2)
        slots = 256 * per_cu
        rounds = -(-p["blocks"] // slots)
        rows.append(dict(name=c.name, plan=p, gflop=c.flops / 1e9, mb=c.bytes / 1e6, fill=p["blocks"] / (rounds * slots)))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--hw", type=int, nargs=2, default=(800, 1344))
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--variant", default="r50", choices=["r50", "ssd", "vis", "benchmark", "dcn"])
    args = ap.parse_args()
    eng = build_on_cpu(args.variant, args.batch, args.hw, args.depth)
    rows = {r["name"]: r for r in conv_rows(eng)}
    print("# %s, batch %d, %dx%d: %d steps, %d conv launches, %.1f conv GFLOP per step" % (
        args.variant, args.batch, args.hw[0], args.hw[1], len(eng.steps), len(eng.convs), sum(r["gflop"] for r in rows.values())))
    print("# lane | step | tile (cout x pos) | K step | loop | blocks | fill of the rounds of resident blocks | GFLOP | algorithmic MB")
    for (label, _), lane in zip(eng.steps, eng.lanes):
        ln = "join " + ",".join(str(x) for x in lane[1:]) if isinstance(lane, tuple) else str(lane)
        if label.startswith("conv:"):
            r = rows[label[5:]]
            p = r["plan"]
            print("%-8s %-34s %3dx%-3d K%-2d %-9s %6d  %3.0f %%  %8.2f %8.1f" % (
                ln, label, p["tile_cout"], p["tile_pos"], p["k_step"],
                {0: "legacy", 1: "flat", 3: "pipelined"}[p["k_loop"]] if p["lds_dma"] else "reg-stage", p["blocks"],
                100 * r["fill"], r["gflop"], r["mb"]))
        elif label != "join":
            print("%-8s %s" % (ln, label))
        else:
            print("%-8s" % ln)


if __name__ == "__main__":
    main()
