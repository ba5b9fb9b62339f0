#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call20
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_train_rows.py -q -m gpu -x > $OUT/pytest_rows.log 2>&1
tail -25 $OUT/pytest_rows.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_api.py -q -m gpu -k "train or loss or sgd or rescoring" > $OUT/pytest_api.log 2>&1
tail -40 $OUT/pytest_api.log | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "training or backward or bwd" > $OUT/pytest_k.log 2>&1
tail -5 $OUT/pytest_k.log | cut -c1-250
timeout 300 python tools/train_bench.py --steps 5 > $OUT/train_rows.json 2>$OUT/train_rows.err
tail -1 $OUT/train_rows.json | cut -c1-400; tail -3 $OUT/train_rows.err | cut -c1-300
