#!/bin/bash
# Round-4 evidence, collected on the GPU box through gpurun from the repo root; tools/collect_profiles4.py then copies it
# into profiles/ under r04_* names:
#   bench lines of every BASELINE config (r50 bf16 with cpu_baseline + parity, head_x3, f32, r101, vis, train, train with the
#   RCCL path forced at world size 1), per-step HIP-event breakdowns (bf16 and x3 chains), rocprofv3 kernel traces of the
#   whole inference step (bf16 and x3) and of the dominant kernel alone, PMC passes on the dominant kernel (FETCH_SIZE /
#   WRITE_SIZE / SQ counters in separate runs, never together with --stats or other trace domains), the parity reports
#   at the BASELINE shape for the three plans.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof4
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --breakdown $OUT/step_breakdown.txt > $OUT/bench_r50.json 2> $OUT/bench_r50.err
timeout 600 python $R/bench.py --precision head_x3 --breakdown $OUT/step_breakdown_x3.txt > $OUT/bench_r50_x3.json 2> $OUT/bench_r50_x3.err
timeout 300 python $R/bench.py --in-flight 1 > $OUT/bench_r50_inflight1.json 2>/dev/null
timeout 600 python $R/bench.py --config r101 > $OUT/bench_r101.json 2> $OUT/bench_r101.err
timeout 600 python $R/bench.py --config vis > $OUT/bench_vis.json 2> $OUT/bench_vis.err
timeout 900 python $R/bench.py --config train --steps 20 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err
SIPMASK_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 1 --config train --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_train_rccl1.json 2> $OUT/bench_train_rccl1.err
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/step -o step -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-graph > $OUT/step.log 2>&1
python $R/tools/prof_stats.py $OUT/step $OUT/kernel_stats_step.csv 5 > /dev/null
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/stepx3 -o stepx3 -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-graph --precision head_x3 > $OUT/stepx3.log 2>&1
python $R/tools/prof_stats.py $OUT/stepx3 $OUT/kernel_stats_step_x3.csv 5 > /dev/null
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/tower -o tower -- python $R/bench.py --tower-only 50 > $OUT/tower.log 2>&1
python $R/tools/prof_stats.py $OUT/tower $OUT/kernel_stats_tower_only.csv 5 > /dev/null
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/towerx3 -o towerx3 -- python $R/bench.py --tower-only 50 --precision head_x3 > $OUT/towerx3.log 2>&1
python $R/tools/prof_stats.py $OUT/towerx3 $OUT/kernel_stats_tower_only_x3.csv 5 > /dev/null
timeout -k 5 600 rocprofv3 --kernel-trace -d $OUT/train -o train -- python $R/bench.py --config train --steps 10 --warmup 3 --no-cpu-baseline > $OUT/train.log 2>&1
python $R/tools/prof_stats.py $OUT/train $OUT/kernel_stats_train_step.csv 5 > /dev/null
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python $R/bench.py --tower-only 10 > $OUT/pmc_$N.log 2>&1
done
rm -rf $OUT/step $OUT/stepx3 $OUT/tower $OUT/towerx3 $OUT/train 2>/dev/null
cd $R
bash tools/marginal_cost.sh $OUT/marginal_cost.txt 1 > /dev/null 2>&1
timeout 600 python tools/parity_baseline.py --plan pipelined --precision bf16 --out $OUT/parity_r50_b4_bf16.json > $OUT/parity_bf16.log 2>&1
timeout 600 python tools/parity_baseline.py --plan pipelined --precision head_x3 --out $OUT/parity_r50_b4_x3.json > $OUT/parity_x3.log 2>&1
find $OUT -name "*counter_collection.csv" | head -3; tail -c 300 $OUT/tower.log; for f in r50 r50_x3 r50_f32 r101 vis train train_rccl1; do cut -c1-200 $OUT/bench_$f.json; done
