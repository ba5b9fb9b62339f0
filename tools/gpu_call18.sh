#!/bin/bash
# row-tensor training graph: op parity, whole-graph parity (head / detector vs oracle autograd), train bench A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call18
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_train_rows.py -q -m gpu > $OUT/pytest_rows.log 2>&1
tail -40 $OUT/pytest_rows.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_api.py -q -m gpu -k "train or loss or sgd" > $OUT/pytest_api.log 2>&1
tail -30 $OUT/pytest_api.log | cut -c1-250
timeout 300 python tools/train_bench.py --steps 5 > $OUT/train_rows.json 2>$OUT/train_rows.err
tail -2 $OUT/train_rows.json | cut -c1-400; tail -5 $OUT/train_rows.err | cut -c1-300
SIPMASK_TRAIN_ROWS=0 timeout 300 python tools/train_bench.py --steps 5 > $OUT/train_nchw.json 2>$OUT/train_nchw.err
tail -1 $OUT/train_nchw.json | cut -c1-400
