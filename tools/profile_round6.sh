#!/bin/bash
# Round-6 evidence, collected on the GPU box through gpurun from the repo root; tools/collect_profiles6.py then copies it
# into profiles/ under r06_* names:
#   the default driver line (bf16, COCO-sized boxes, parity_pairs, post_processing, other_configs with cpu baselines), the same
#   line with the raw synthetic boxes (--det-boxes tiny: comparable with rounds 1-4), head_x3 with its breakdown, eval_shapes,
#   per-step HIP-event breakdowns, rocprofv3 kernel traces of the inference step (bf16 / x3), of the dominant kernel alone and of the
#   training step, PMC passes on the dominant kernel (FETCH_SIZE / WRITE_SIZE / SQ counters in separate runs, never together with
#   --stats or other trace domains), parity reports at the BASELINE shape, microbenchmarks of the kernels written this round.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof6
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $OUT/bench_r50_driver_args.json 2> $OUT/bench_r50_driver_args.err
timeout 300 python $R/bench.py --no-cpu-baseline --extras-budget 45 --breakdown $OUT/step_breakdown.txt > $OUT/bench_r50.json 2> $OUT/bench_r50.err
timeout 300 python $R/bench.py --no-cpu-baseline --extras-budget 45 --det-boxes tiny > $OUT/bench_r50_tiny_boxes.json 2> $OUT/bench_r50_tiny.err
timeout 300 python $R/bench.py --precision head_x3 --no-cpu-baseline --extras-budget 30 --breakdown $OUT/step_breakdown_x3.txt > $OUT/bench_r50_x3.json 2> $OUT/bench_r50_x3.err
timeout 300 python $R/bench.py --config eval_shapes > $OUT/bench_eval_shapes.json 2> $OUT/bench_eval_shapes.err
timeout 300 python $R/bench.py --config ssd --cpu-budget 8 --breakdown $OUT/step_breakdown_ssd.txt > $OUT/bench_ssd.json 2> $OUT/bench_ssd.err
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/step -o step -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $OUT/step.log 2>&1
python $R/tools/prof_stats.py $OUT/step $OUT/kernel_stats_step.csv 5 > /dev/null
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/stepx3 -o stepx3 -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-graph --precision head_x3 > $OUT/stepx3.log 2>&1
python $R/tools/prof_stats.py $OUT/stepx3 $OUT/kernel_stats_step_x3.csv 5 > /dev/null
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/tower -o tower -- python $R/bench.py --tower-only 50 > $OUT/tower.log 2>&1
python $R/tools/prof_stats.py $OUT/tower $OUT/kernel_stats_tower_only.csv 5 > /dev/null
timeout -k 5 600 rocprofv3 --kernel-trace -d $OUT/train -o train -- python $R/bench.py --config train --steps 10 --warmup 3 --no-cpu-baseline > $OUT/train.log 2>&1
python $R/tools/prof_stats.py $OUT/train $OUT/kernel_stats_train_step.csv 5 > /dev/null
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python $R/bench.py --tower-only 10 > $OUT/pmc_$N.log 2>&1
done
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/towerx3 -o towerx3 -- python $R/bench.py --precision head_x3 --tower-only 50 > $OUT/tower_x3.log 2>&1
python $R/tools/prof_stats.py $OUT/towerx3 $OUT/kernel_stats_tower_only_x3.csv 5 > /dev/null
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmcx3_$N -- python $R/bench.py --precision head_x3 --tower-only 10 > $OUT/pmcx3_$N.log 2>&1
done
rm -rf $OUT/step $OUT/stepx3 $OUT/tower $OUT/towerx3 $OUT/train 2>/dev/null
cd $R
timeout 300 python tools/deform_fwd_bench.py 4 0.0,1.0,2.0,4.0 > $OUT/deform_fwd_microbench.txt 2>&1
timeout 300 python tools/deform_bwd_bench.py > $OUT/deform_bwd_microbench.txt 2>&1
bash tools/marginal_cost_x3.sh $OUT/marginal_cost_x3.txt 1 > /dev/null 2>&1
bash tools/marginal_cost.sh $OUT/marginal_cost_bf16.txt 1 > /dev/null 2>&1
timeout 600 python tools/parity_baseline.py --plan pipelined --precision bf16 --out $OUT/parity_r50_b4_bf16.json > $OUT/parity_bf16.log 2>&1
timeout 600 python tools/parity_baseline.py --plan pipelined --precision head_x3 --out $OUT/parity_r50_b4_x3.json > $OUT/parity_x3.log 2>&1
find $OUT -name "*counter_collection.csv" | head -3; tail -c 300 $OUT/tower.log; for f in r50_driver_args r50 r50_tiny_boxes r50_x3 eval_shapes ssd; do cut -c1-160 $OUT/bench_$f.json; done
