#!/bin/bash
# round 3, call 25: SQ counters of wgrad_direct_kernel on the tower weight gradient
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/wgpmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-20)
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$N -- python $R/tools/wgrad_tower_loop.py > $OUT/$N.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, statistics
acc = collections.defaultdict(list); dur = []
for f in glob.glob("gpurun_out/wgpmc/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_direct" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/wgpmc/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_direct" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("wgrad_direct_kernel<2,4,4,2>, tower dW (9 taps x 57 slices = 513 blocks of 512 threads), per launch:")
print("duration us: mean %.1f (n=%d)" % (statistics.mean(dur), len(dur)))
for k in sorted(acc): print("%-32s %.4g" % (k, statistics.mean(acc[k])))
PY
