#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench line (with cpu_baseline), per-step HIP-event breakdown, rocprofv3 kernel stats of the whole step and
#   of the dominant kernel alone, and the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 20 --warmup 5 --breakdown $OUT/step_breakdown.txt > $OUT/bench_full.json 2> $OUT/bench_full.err
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step -- \
  python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $OUT/step.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tower -- \
  python $R/bench.py --tower-only 50 > $OUT/tower.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- \
    python $R/bench.py --tower-only 10 > $OUT/pmc_$C.log 2>&1
done
python $R/tools/conv_bench.py --variants 0 --rounds 3 > $OUT/conv_microbench.txt 2>&1
find $OUT -name "*.csv" | head -40
tail -2 $OUT/bench_full.json
