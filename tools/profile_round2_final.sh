#!/bin/bash
# Round-2 FINAL-build evidence (supersedes tools/profile_round2.sh's mid-round set), collected on the GPU box through gpurun
# from the repo root; tools/collect_profiles2.py then copies it into profiles/ under the r02_* names:
#   bench lines (r50 with cpu_baseline, --lanes 1, f32 plan, r101, vis, train), per-step HIP-event breakdowns, rocprofv3
#   kernel traces of the whole inference step, of the dominant kernel alone and of the training step, PMC passes on the
#   dominant kernel (FETCH_SIZE / WRITE_SIZE / SQ counters in separate runs, never together with --stats or other trace
#   domains), the parity reports at the BASELINE shape, the patch-conv micro-benchmarks.
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
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o train -- python $R/tools/train_bench.py --steps 2 > $OUT/train.log 2>&1
cp $(find $OUT/train -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_train.csv
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python $R/bench.py --tower-only 10 > $OUT/pmc_$N.log 2>&1
done
rm -rf $OUT/step $OUT/tower $OUT/train 2>/dev/null
cd $R
timeout 600 python tools/parity_baseline.py --depth 50 --batch 4 --precision bf16 --out $OUT/parity_r50_b4_bf16.json > $OUT/parity_bf16.log 2>&1
timeout 600 python tools/parity_baseline.py --depth 50 --batch 4 --precision f32 --out $OUT/parity_r50_b4_f32.json > $OUT/parity_f32.log 2>&1
timeout 300 python tools/patch_bench.py > $OUT/patch_bench.txt 2>&1
timeout 300 python tools/deform_fwd_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/deform_bench.txt
find $OUT -name "*counter_collection.csv" | head -3; tail -c 300 $OUT/tower.log; cut -c1-250 $OUT/bench_r50.json; cut -c1-160 $OUT/bench_train.json
