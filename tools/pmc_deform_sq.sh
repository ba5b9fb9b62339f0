cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_dsq; rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python $R/tools/deform_fwd_bench.py 2 0.3 > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv,glob,collections,os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_dsq"
for f in sorted(glob.glob(root+"/*/*/*_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "deform_patch_kernel" in k:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(k, len(v), sum(v)/len(v))
PY
