# LDS bank-conflict counters of the two LDS-patch kernels (separate --pmc pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_lds2; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/deform -- python $R/tools/deform_fwd_bench.py > $OUT/deform.log 2>&1
timeout -k 5 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/patch -- python $R/bench.py --tower-only 5 > $OUT/patch.log 2>&1
python - <<'PY'
import csv,glob,collections,os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_lds2"
for f in sorted(glob.glob(root+"/*/*/*_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "deform_patch_kernel" in k or "conv3x3_patch_kernel" in k or ("conv_igemm_kernel" in k and "true, false" in k):
            acc[(k[:60], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(k, len(v), sum(v)/len(v))
PY
