#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/wgpmc2; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-20)
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$N -- python $R/tools/wgrad_tower_loop.py > $OUT/$N.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, statistics
acc = collections.defaultdict(list); dur = []
for f in glob.glob("gpurun_out/wgpmc2/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_direct" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/wgpmc2/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_direct" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("duration us: mean %.1f (n=%d)" % (statistics.mean(dur), len(dur)))
for k in sorted(acc): print("%-32s %.4g" % (k, statistics.mean(acc[k])))
PY
