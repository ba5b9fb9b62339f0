#!/usr/bin/env python
"""Rough CU-time share per stage of one SubBatchPlan chain: the per-launch HIP-event times of the eager chain
(profiles/r02_step_breakdown_hip_events.txt) weighted by the fraction of the chip each launch can occupy (blocks / resident
slots from the host-side launch plan, profiles/r02_launch_plan_r50_subplan.txt; element-wise kernels = 1, the few-block
post-processing kernels = 0.05).  The benchmarked step is bound by aggregate CU time (two chains fill each other's gaps:
free-running chains +1 %), so this -- not the chain's latency -- ranks what to attack next.  No GPU."""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# usage: cu_time_model.py [breakdown.txt launch_plan.txt chains_in_flight]   (defaults: the round-2 sub-plan chain, 2)
BREAKDOWN = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_step_breakdown_hip_events.txt")
PLAN = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r02_launch_plan_r50_subplan.txt")
CHAINS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ms = {}
for l in open(BREAKDOWN):
    m = re.match(r"(\S+)\s+([\d.]+) ms", l)
    if m:
        ms.setdefault(m.group(1), []).append(float(m.group(2)))
occ = {}
for l in open(PLAN):
    p = l.split()
    if len(p) > 6 and p[1].startswith("conv:") and p[2] in ("igemm", "patch", "window"):
        occ[p[1]] = min(1.0, float(p[5]))


def stage(n):
    if "layer" in n:
        return n.split(".")[1]
    if "stem" in n or n in ("nhwc", "maxpool"):
        return "stem"
    if "fpn" in n or n == "relu:p6":
        return "fpn"
    if "tower" in n or "reg_convs.3" in n or n.startswith("gn:cls") or n.startswith("gn:reg"):
        return "towers+GN"
    if n in ("det_select", "nms", "mask_assemble"):
        return "post"
    if "feat_align" in n or n == "offset":
        return "feat_align"
    return "head other"


tot, cut = collections.OrderedDict(), collections.OrderedDict()
for n, v in ms.items():
    if n in ("join", "#"):
        continue
    t, s = sum(v), stage(n)
    elementwise = n.startswith(("gn:", "up:", "nhwc", "maxpool", "relu"))
    o = occ.get(n, 1.0 if elementwise else (0.05 if n in ("nms", "det_select") else (0.8 if "tail" in n else 0.5)))
    tot[s] = tot.get(s, 0) + t
    cut[s] = cut.get(s, 0) + t * o
T, Cs = sum(tot.values()), sum(cut.values())
print("stage          ms (chain alone)  share    CU-time (ms x occupied fraction)  share")
for s in tot:
    print("%-14s %8.3f %8.1f %% %14.3f %18.1f %%" % (s, tot[s], 100 * tot[s] / T, cut[s], 100 * cut[s] / Cs))
print("sum %.3f ms; CU-time %.3f ms per chain; %d chain(s) per step -> %.2f ms of chip time per step" % (T, Cs, CHAINS if CHAINS == 2 and "r02" in BREAKDOWN else 1, (CHAINS if CHAINS == 2 and "r02" in BREAKDOWN else 1) * Cs))
