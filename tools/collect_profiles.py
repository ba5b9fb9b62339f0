#!/usr/bin/env python
"""Copies the evidence written by tools/profile_round.sh (gpurun_out/prof) into profiles/ with the round's names and
derives profiles/r01_pmc_tower_conv.json (per-launch HBM traffic of the dominant kernel: FETCH_SIZE x2 on gfx950 +
WRITE_SIZE, dispatches selected by kernel name and the tower launch's grid size)."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01c"
KNAME = "conv_igemm_kernel<2, 2, 2, 2, false, true"
GRID = "359424"


def one(pattern):
    # gpurun_out/ accumulates the files of every call: take the newest match
    return max(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)


def with_header(src, dst, header):
    with open(dst, "w") as f:
        f.write("# " + header + "\n")
        f.write(open(src).read())


with_header(one("step/*/*_kernel_stats.csv"), os.path.join(DST, TAG + "_rocprofv3_kernel_stats_step.csv"),
            "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "
            "--no-graph   (tools/profile_round.sh; whole plan incl. calibration / warm-up / breakdown passes)")
with_header(one("tower/*/*_kernel_stats.csv"), os.path.join(DST, TAG + "_rocprofv3_kernel_stats_tower_only.csv"),
            "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --tower-only 50   (dominant kernel "
            "alone; the tower-shaped dispatches have grid 359424, their trace statistics are in r01_pmc_tower_conv.json)")
shutil.copy(os.path.join(SRC, "step_breakdown.txt"), os.path.join(DST, "r01_step_breakdown_hip_events.txt"))
if os.path.getmtime(os.path.join(SRC, "conv_microbench.txt")) >= os.path.getmtime(os.path.join(SRC, "bench_full.json")):   # same call only
    shutil.copy(os.path.join(SRC, "conv_microbench.txt"), os.path.join(DST, "r01_conv_microbench.txt"))
line = [l for l in open(os.path.join(SRC, "bench_full.json")) if l.startswith("{")][-1]
open(os.path.join(DST, "r01_bench_line.json"), "w").write(line)
bench = json.loads(line)

cnt = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [r for r in csv.DictReader(open(one("pmc_%s/*/*_counter_collection.csv" % c)))
            if KNAME in r["Kernel_Name"] and r["Grid_Size"] == GRID and r["Counter_Name"] == c]
    v = [float(r["Counter_Value"]) for r in rows]
    cnt[c] = (len(v), statistics.mean(v))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(one("tower/*/*_kernel_trace.csv")))
     if KNAME in r["Kernel_Name"] and r["Grid_Size_X"] == GRID]
tower_live = [l for l in open(os.path.join(SRC, "tower.log")) if l.startswith("{")]
old = json.load(open(os.path.join(DST, "r01_pmc_tower_conv.json")))
fetch, write = cnt["FETCH_SIZE"][1] * 1024 * 2, cnt["WRITE_SIZE"][1] * 1024
js = {
    "kernel": "conv_igemm_kernel<2,2,2,2,false,true,0,3> with fused GroupNorm statistics = tower 3x3 256->256 over 5 FPN "
              "levels, B=4 (M=89600,N=256,K=2304), grid 1404 x 256 threads",
    "command": "rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python bench.py --tower-only 10   "
               "(tools/profile_round.sh; separate passes for FETCH_SIZE and WRITE_SIZE; dispatches selected by kernel "
               "name + grid size 359424; assembled by tools/collect_profiles.py)",
    "FETCH_SIZE_KB_raw": cnt["FETCH_SIZE"][1], "WRITE_SIZE_KB_raw": cnt["WRITE_SIZE"][1], "dispatches": cnt["FETCH_SIZE"][0],
    "fetch_bytes_corrected": fetch, "write_bytes": write,
    "correction": old["correction"],
    "hbm_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": 92900000.0,
    "rocprofv3_kernel_trace_ns": {"n": len(d), "mean": statistics.mean(d), "median": statistics.median(d), "min": min(d),
                                  "max": max(d)},
    "live_hip_event_ms_per_launch": {"bench.py (no profiler, eager breakdown)": bench["roofline"]["ms_per_launch"],
                                     "bench.py --tower-only 50 under rocprofv3": json.loads(tower_live[-1])["ms_per_launch"]
                                     if tower_live else None},
    "SQ_first_build": old.get("SQ_first_build"),
}
json.dump(js, open(os.path.join(DST, "r01_pmc_tower_conv.json"), "w"), indent=1)
print(json.dumps({k: js[k] for k in ("hbm_bytes_per_launch", "rocprofv3_kernel_trace_ns", "live_hip_event_ms_per_launch")}))
