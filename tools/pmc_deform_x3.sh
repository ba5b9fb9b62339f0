# PMC passes on the x3 FeatureAlign window kernel (csrc/deform_patch_x3.hip): B=4 head, offsets ~N(0, 1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_dx3; rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python $R/tools/deform_fwd_bench.py 4 1.0 > $OUT/p$i.log 2>&1
done
python - <<'PY' > $R/gpurun_out/pmc_dx3_summary.txt
import csv,glob,collections,os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_dx3"
for f in sorted(glob.glob(root+"/*/*/*_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "deform_patch_x3" in k:
            tag = "x3_window"
        elif "deform_patch_kernel" in k:
            tag = "bf16_patch"
        elif "conv_f32_kernel" in k:
            tag = "x3_gather"
        else:
            continue
        acc[(tag, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(k[0], k[1], len(v), sum(v)/len(v))
PY
cat $R/gpurun_out/pmc_dx3_summary.txt
rm -rf $OUT
