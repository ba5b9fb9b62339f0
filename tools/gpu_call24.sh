#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call24
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $R/tools/train_bench.py --steps 2 > $OUT/prof_train.log 2>&1
cd $R
cp $(find $OUT/prof_train -name "*kernel_stats.csv" | head -1) $OUT/train_kernel_stats.csv
find $OUT/prof_train -name "*kernel_trace.csv" -delete
