#!/usr/bin/env python
"""Prints the static launch plan of an engine configuration WITHOUT a GPU: every conv launch with the tile, K-step
width, K-loop variant and grid that `sm_conv_plan_query` (the launcher's own selection code) picks for it, the
fill of its last round of resident blocks, and the non-conv steps with their lanes.

    python tools/plan_dump.py [--batch 4] [--hw 800 1344] [--depth 50] [--variant r50|ssd|vis|benchmark|dcn] [--sub-plan]

--sub-plan: the plan of ONE chain of engine.SubBatchPlan (the benchmarked structure: --batch 2 --sub-plan): no split-K,
uniform patch-conv launches.  Round 2: the LDS-window kernels (conv3x3_patch.hip, deform_patch.hip) and the fused
bottleneck tails (bottleneck.hip) are listed with their own launch shapes (sm_conv3x3_patch_plan,
sm_deform_conv_window_plan -- host logic as well).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_on_cpu(variant="r50", batch=4, hw=(800, 1344), depth=50, sub_plan=False, precision="bf16", pipelined=False):
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
        return SipMaskEngine(sd, batch, tuple(hw), depth, device="cpu", sub_plan=sub_plan, precision=precision,
                             pipelined=pipelined, **kw)
    finally:
        torch.cuda.is_available = real


def conv_rows(eng):
    from sipmask_amd import hip_ops as H
    rows = []
    for c in eng.convs:
        if getattr(c, "f32", False):                     # conv_f32.hip (the exact-f32 MFMA kernel): 128x128 tiles
            blocks = sum(-(-c.desc.batch * c.desc.out_h[l] * c.desc.out_w[l] // 128) for l in range(c.desc.nlev)) * (c.desc.cout_pad // 128 or 1)
            rows.append(dict(name=c.name, kind="f32", shape="128x128", blocks=blocks, waves=blocks / 512.0,
                             note="exact-f32 MFMA%s" % (" (deformable)" if c.offset is not None else ""),
                             gflop=c.flops / 1e9, mb=c.bytes / 1e6))
            continue
        if getattr(c, "smallco", False):                 # conv3x3_smallco.hip: one wave per 2 x 32-position tile
            blocks = H.conv3x3_smallco_tiles(c.desc)
            rows.append(dict(name=c.name, kind="smallco", shape="32x(2x32)", blocks=blocks, waves=blocks / (256 * 5.0),
                             note="one-wave tiles, weights as MFMA fragments from L2", gflop=c.flops / 1e9, mb=c.bytes / 1e6))
            continue
        if getattr(c, "patch", False):                   # conv3x3_patch.hip: one block per CU, its own launch planner
            pp = H.conv3x3_patch_plan(c.desc)
            blocks = pp["big"] + pp["small"]
            bco = 128 if c.desc.patch_cout_tile == 128 else (32 if c.desc.cout_pad == 32 else 256)    # cout tile of the launch
            shape = "%dx256" % bco + ("" if not pp["small"] else "+%d" % pp["small_pos"])
            rows.append(dict(name=c.name, kind="patch", shape=shape, blocks=blocks, waves=blocks / 256.0,
                             note="LDS patch, makespan %.2f tiles, CU fill %.0f %%" % (pp["makespan"], 100 * pp["fill"]),
                             gflop=c.flops / 1e9, mb=c.bytes / 1e6))
            continue
        if c.offset is not None:
            wp = H.deform_conv_window_plan(c.desc)
            if wp is not None:                            # deform_patch.hip
                rows.append(dict(name=c.name, kind="window", shape="256x(%dx%d)" % wp["tile"], blocks=wp["blocks"],
                                 waves=wp["blocks"] / 256.0, note="LDS window %d px / deformable group" % wp["window_pixels"],
                                 gflop=c.flops / 1e9, mb=c.bytes / 1e6))
                continue
        p = H.conv_plan(c.desc, deformable=c.offset is not None, with_gn_stats=c.gn_stats is not None)
        if p.get("split_k", 1) > 1 and getattr(c, "ws", None) is None:
            # the engine gave this conv no split-K workspace (sub-plan chains, lanes): the launcher then plans without the split
            import copy
            d2 = copy.copy(c.desc)
            d2.flags |= 0x00010000                       # SM_CONV_DBG_NO_SPLITK
            p = H.conv_plan(d2, deformable=c.offset is not None, with_gn_stats=c.gn_stats is not None)
        per_cu = 1 if p["threads"] == 512 else (4 if p["k_step"] == 32 else 2)
        slots = 256 * per_cu
        # waves = blocks / resident slots: with several blocks per CU the leftover blocks of a barely started round run
        # alone and faster, so a low fill costs less than its face value
        loop = {0: "legacy", 1: "flat", 3: "pipelined"}[p["k_loop"]] if p["lds_dma"] else "reg-stage"
        rows.append(dict(name=c.name, kind="igemm", plan=p, shape="%dx%d" % (p["tile_cout"], p["tile_pos"]), blocks=p["blocks"],
                         waves=p["blocks"] / slots, note="K%d %s%s" % (p["k_step"], loop, "" if p.get("split_k", 1) <= 1
                                                                      else " split-K %d" % p["split_k"]),
                         gflop=c.flops / 1e9, mb=c.bytes / 1e6))
    for t in eng.fused:                                   # bottleneck.hip: conv2 + conv3 (+ next conv1) per launch
        rows.append(dict(name=t.name, kind="fused", shape="tail", blocks=0, waves=0.0,
                         note=("conv3+next conv1 in one launch" if getattr(t, "convs_in_launch", 0) == 2 and not hasattr(t, "w2") else
                               "conv2+conv3%s%s in one launch" % ("+shortcut conv" if t.x_block is not None else "",
                                                                  "+next conv1" if t.w1n is not None else "")),
                         gflop=t.flops / 1e9, mb=t.bytes / 1e6))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--hw", type=int, nargs=2, default=(800, 1344))
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--variant", default="r50", choices=["r50", "ssd", "vis", "benchmark", "dcn"])
    ap.add_argument("--sub-plan", action="store_true")
    ap.add_argument("--pipelined", action="store_true", help="one slot of a PipelinedPlan (big tiles, no split-K, no side lanes)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32", "head_x3"])
    args = ap.parse_args()
    eng = build_on_cpu(args.variant, args.batch, args.hw, args.depth, args.sub_plan, args.precision, args.pipelined)
    rows = {r["name"]: r for r in conv_rows(eng)}
    print("# %s, batch %d%s, %dx%d: %d steps, %d conv launches (+ %d fused bottleneck tails), %.1f conv GFLOP per step" % (
        args.variant, args.batch, " (one chain of a SubBatchPlan)" if args.sub_plan else (" (one slot of a PipelinedPlan)" if args.pipelined else ""), args.hw[0], args.hw[1],
        len(eng.steps), len(eng.convs), len(eng.fused), eng.total_conv_flops() / 1e9))
    print("# lane | step | kernel | tile (cout x pos) | blocks | blocks / resident slots (256 CUs x 1, 2 or 4) | GFLOP | "
          "algorithmic MB | notes")
    for (label, _), lane in zip(eng.steps, eng.lanes):
        ln = "join " + ",".join(str(x) for x in lane[1:]) if isinstance(lane, tuple) else str(lane)
        if label.startswith("conv:") and label[5:] in rows:
            r = rows[label[5:]]
            print("%-8s %-36s %-6s %-12s %6d  %5.2f  %8.2f %8.1f  %s" % (
                ln, label, r["kind"], r["shape"], r["blocks"], r["waves"], r["gflop"], r["mb"], r["note"]))
        elif label != "join":
            print("%-8s %s" % (ln, label))
        else:
            print("%-8s" % ln)


if __name__ == "__main__":
    main()
