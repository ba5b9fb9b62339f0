#!/bin/bash
# session re-entry: confirm HEAD on a box (bench default), kernel-level profile of the training step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call14
rm -rf $OUT; mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_r50.json 2>$OUT/bench_r50.err
tail -1 $OUT/bench_r50.json | cut -c1-300
timeout 300 python bench.py --config train --no-cpu-baseline > $OUT/bench_train.json 2>$OUT/bench_train.err
tail -1 $OUT/bench_train.json | cut -c1-300
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_train -o train -- python $R/tools/train_bench.py --steps 2 > $OUT/prof_train.log 2>&1
cd $R
find $OUT/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} head -45 {}
find $OUT/prof_train -name "*kernel_trace.csv" -delete
find $OUT/prof_train -name "*.db" -delete
timeout 200 python tools/prof_train_host.py > $OUT/host_prof.txt 2>&1
tail -40 $OUT/host_prof.txt
