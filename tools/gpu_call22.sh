#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call22
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_api.py -q -m gpu -k "loss or train" > $OUT/pytest_api.log 2>&1
tail -5 $OUT/pytest_api.log | cut -c1-250
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $R/tools/train_bench.py --steps 2 > $OUT/prof_train.log 2>&1
cd $R
F=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1)
cp $F $OUT/train_kernel_stats.csv
find $OUT/prof_train -name "*kernel_trace.csv" -delete
timeout 300 python tools/train_bench.py --steps 5 > $OUT/train_rows.json 2>$OUT/train_rows.err
tail -1 $OUT/train_rows.json | cut -c1-300
