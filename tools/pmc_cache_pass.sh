# L1 <-> L2 request counters of the benchmarked grouped tower launch (conv3x3_patch_kernel, grid 188416): is the 64-byte-row
# LDS-DMA paying for whole 128-byte lines?  Separate --pmc passes, kernel trace only (no other trace domain).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_cache; mkdir -p $OUT
for C in "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$N -- python $R/bench.py --tower-only 5 > $OUT/$N.log 2>&1
  tail -2 $OUT/$N.log
done
python - <<'PY'
import csv,glob,collections,os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_cache"
for f in sorted(glob.glob(root+"/*/*/*_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv3x3_patch_kernel" in r["Kernel_Name"] and r["Grid_Size"]=="188416":
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
