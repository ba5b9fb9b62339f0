#!/usr/bin/env python
"""Copies the evidence written by tools/profile_round6.sh (gpurun_out/prof6) into profiles/ under round-6 names and
derives profiles/r06_pmc_tower_conv.json: per-launch HBM traffic (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate PMC
passes) and the SQ counters of the dominant kernel, dispatches selected by kernel name and the launch's grid."""
import collections
import csv
import glob
import json
import os
import shutil
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof6")
DST = os.path.join(ROOT, "profiles")


def last_json(path):
    if not os.path.exists(path):
        return None
    ls = [l for l in open(path) if l.startswith("{")]
    return json.loads(ls[-1]) if ls else None


def pmc_report(tower_log, pmc_prefix, stats_csv, bench_json, out_name):
    tower = last_json(os.path.join(SRC, tower_log))
    if tower is None:
        return None
    kname = "conv3x3_patch_kernel" if tower["patch_kernel"] else "conv_dma32_kernel"

    def counters(pattern):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(SRC, pattern, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kname in r["Kernel_Name"]:
                    acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return acc

    def mean(v):
        return statistics.mean(v) if v else None

    fetch = counters(pmc_prefix + "_FETCH_SIZE")
    if not fetch:
        return None
    grid = max(fetch, key=lambda g: len(fetch[g]["FETCH_SIZE"]))          # the launch --tower-only repeats

    def trace_by_grid():
        out = {}
        for f in sorted(glob.glob(os.path.join(SRC, pmc_prefix + "_*", "**", "*_kernel_trace.csv"), recursive=True)):
            v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f))
                 if kname in r["Kernel_Name"] and r["Grid_Size_X"] == grid]
            if v:
                out[f.split(os.sep)[-3]] = {"calls": len(v), "avg": round(mean(v), 2), "min": round(min(v), 2), "max": round(max(v), 2)}
        return out

    write, sq = counters(pmc_prefix + "_WRITE_SIZE"), counters(pmc_prefix + "_SQ_VALU_MFMA_BUSY_CYCLES")
    fb, wb = mean(fetch[grid]["FETCH_SIZE"]) * 1024 * 2, mean(write[grid]["WRITE_SIZE"]) * 1024
    bench = last_json(os.path.join(SRC, bench_json))
    stats = list(csv.DictReader(open(os.path.join(SRC, stats_csv)))) if os.path.exists(os.path.join(SRC, stats_csv)) else []
    krow = ([r for r in stats if kname in "".join(str(v) for v in r.values())] or [None])[0]
    js = {
        "kernel": bench["roofline"].get("kernel") if bench else None,
        "launch": tower["kernel"], "mode": tower.get("mode"), "plan_batch": tower["plan_batch"], "grid_size": int(grid),
        "command": "rocprofv3 --kernel-trace --pmc <COUNTER(S)> --output-format csv -- python bench.py [--precision head_x3] "
                   "--tower-only 10 (tools/profile_round6.sh; separate passes for FETCH_SIZE, WRITE_SIZE and the SQ counters; "
                   "dispatches selected by kernel name + grid size; assembled by tools/collect_profiles6.py)",
        "dispatches": len(fetch[grid]["FETCH_SIZE"]),
        "FETCH_SIZE_KB_raw": mean(fetch[grid]["FETCH_SIZE"]), "WRITE_SIZE_KB_raw": mean(write[grid]["WRITE_SIZE"]),
        "fetch_bytes_corrected": fb, "write_bytes": wb,
        "correction": "gfx950 rocprofv3 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM "
                      "section): doubled; WRITE_SIZE used as is; both in KB",
        "hbm_bytes_per_launch": fb + wb, "algorithmic_bytes_per_launch": tower["algorithmic_mb"] * 1e6,
        "gflop_per_launch": tower.get("gflop"), "tower_only_ms_per_launch_under_profiler": tower["ms_per_launch"],
        "rocprofv3_kernel_trace": krow,
        "live_hip_event_ms_per_launch": {"bench.py breakdown (no profiler)": bench["roofline"].get("ms_per_launch") if bench else None},
        "SQ_per_dispatch": {k: mean(v) for k, v in sq[grid].items()},
        "rocprofv3_kernel_trace_this_grid_us": trace_by_grid(),
    }
    json.dump(js, open(os.path.join(DST, out_name), "w"), indent=1)
    return js


js = pmc_report("tower.log", "pmc", "kernel_stats_tower_only.csv", "bench_r50.json", "r06_pmc_tower_conv.json")
jx = pmc_report("tower_x3.log", "pmcx3", "kernel_stats_tower_only_x3.csv", "bench_r50_x3.json", "r06_pmc_tower_conv_head_x3.json")
for src, dst in (("step_breakdown.txt", "r06_step_breakdown_hip_events.txt"),
                 ("step_breakdown_x3.txt", "r06_step_breakdown_hip_events_head_x3.txt"),
                 ("step_breakdown_ssd.txt", "r06_step_breakdown_hip_events_ssd544.txt"),
                 ("kernel_stats_step.csv", "r06_rocprofv3_kernel_stats_step.csv"),
                 ("kernel_stats_step_x3.csv", "r06_rocprofv3_kernel_stats_step_head_x3.csv"),
                 ("kernel_stats_tower_only.csv", "r06_rocprofv3_kernel_stats_tower_only.csv"),
                 ("kernel_stats_train_step.csv", "r06_rocprofv3_kernel_stats_train_step.csv"),
                 ("kernel_stats_tower_only_x3.csv", "r06_rocprofv3_kernel_stats_tower_only_head_x3.csv"),
                 ("parity_r50_b4_bf16.json", "r06_parity_r50_b4_bf16_pipelined.json"),
                 ("parity_r50_b4_x3.json", "r06_parity_r50_b4_head_x3_pipelined.json"),
                 ("deform_fwd_microbench.txt", "r06_deform_conv_microbench.txt"),
                 ("deform_bwd_microbench.txt", "r06_deform_bwd_gather_vs_scatter.txt"),
                 ("marginal_cost_x3.txt", "r06_marginal_cost_pipelined_step_head_x3.txt"),
                 ("marginal_cost_bf16.txt", "r06_marginal_cost_pipelined_step.txt")):
    if os.path.exists(os.path.join(SRC, src)):
        shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))
lines = {}
for n in ("r50_driver_args", "r50", "r50_tiny_boxes", "r50_x3", "eval_shapes", "ssd"):
    j = last_json(os.path.join(SRC, "bench_%s.json" % n))
    if j:
        lines[n] = j
json.dump(lines, open(os.path.join(DST, "r06_bench_lines.json"), "w"), indent=1)
print(json.dumps(js, indent=1)[:1500]); print(json.dumps(jx, indent=1)[:1500] if jx else None)
if lines.get("ssd"):
    json.dump(lines["ssd"], open(os.path.join(DST, "r06_bench_line_ssd544.json"), "w"), indent=1)
dl = lines.get("r50_driver_args")
if dl:
    json.dump(dl, open(os.path.join(DST, "r06_bench_line_driver_args.json"), "w"), indent=1)
print({k: (v["value"], v["unit"]) for k, v in lines.items()})
