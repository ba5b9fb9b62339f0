#!/bin/bash
# Round-2 evidence, collected on the GPU box through gpurun from the repo root:
#   bench lines (r50 with cpu_baseline, f32 plan, r101, vis, train), the per-step HIP-event breakdown, rocprofv3 kernel
#   traces of the whole step and of the dominant kernel alone, and PMC passes on the dominant kernel (HBM traffic:
#   FETCH_SIZE / WRITE_SIZE in separate runs; MFMA busy) -- counters never together with --stats / other trace domains.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --breakdown $OUT/step_breakdown.txt > $OUT/bench_r50.json 2> $OUT/bench_r50.err
timeout 300 python $R/bench.py --no-cpu-baseline --lanes 1 --breakdown $OUT/step_breakdown_lanes1.txt > $OUT/bench_r50_lanes1.json 2>/dev/null
timeout 300 python $R/bench.py --no-cpu-baseline --precision f32 > $OUT/bench_r50_f32.json 2>/dev/null
timeout 300 python $R/bench.py --no-cpu-baseline --config r101 > $OUT/bench_r101.json 2>/dev/null
timeout 300 python $R/bench.py --config vis > $OUT/bench_vis.json 2>/dev/null
timeout 300 python $R/bench.py --config train > $OUT/bench_train.json 2>/dev/null
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/step -o step -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-graph > $OUT/step.log 2>&1
python $R/tools/prof_stats.py $OUT/step $OUT/kernel_stats_step.csv 5 > /dev/null
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/tower -o tower -- python $R/bench.py --tower-only 50 > $OUT/tower.log 2>&1
python $R/tools/prof_stats.py $OUT/tower $OUT/kernel_stats_tower_only.csv 5 > /dev/null
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python $R/bench.py --tower-only 10 > $OUT/pmc_$N.log 2>&1
done
rm -rf $OUT/step/*.db $OUT/tower/*.db $OUT/step $OUT/tower 2>/dev/null
find $OUT -name "*counter_collection.csv" | head; tail -c 300 $OUT/tower.log; cut -c1-250 $OUT/bench_r50.json
