cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc2; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_INSTS_LDS\|SQ_INST_CYCLES_VMEM[A-Z_]*\|TCP_PENDING_STALL_CYCLES\|TCP_TCC_READ_REQ_sum\|TCC_HIT_sum\|TCC_MISS_sum\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*\|SQ_WAIT_INST_LDS\|SQ_BUSY_CU_CYCLES\|TCP_TA_TCP_STATE_READ\|TA_BUSY_avr\|TCP_READ_TAGCONFLICT_STALL_CYCLES_sum\|TCP_TCR_TCP_STALL_CYCLES_sum" | sort -u > $OUT/avail.txt
cat $OUT/avail.txt | tr '\n' ' '
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$N -- python $R/bench.py --tower-only 5 > $OUT/$N.log 2>&1
done
python - <<'PY'
import csv,glob,collections,os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc2"
for f in sorted(glob.glob(root+"/*/*/*_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_igemm_kernel<2, 2, 2, 2, false, true>" in r["Kernel_Name"] and r["Grid_Size"]=="359424":
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
