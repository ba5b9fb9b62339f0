#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call21
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_benchmark_variant.py -q -m gpu > $OUT/pytest_b.log 2>&1
tail -40 $OUT/pytest_b.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_api.py -q -m gpu -k "detector_forward_train" > $OUT/pytest_api.log 2>&1
tail -5 $OUT/pytest_api.log | cut -c1-250
timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_r50.json 2>$OUT/bench_r50.err
tail -1 $OUT/bench_r50.json | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --lanes 1 > $OUT/bench_r50_l1.json 2>$OUT/bench_r50_l1.err
tail -1 $OUT/bench_r50_l1.json | cut -c1-200
SIPMASK_CONV_DEBUG_FLAGS=0x4000 timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --lanes 1 > $OUT/bench_r50_l1u.json 2>$OUT/bench_r50_l1u.err
tail -1 $OUT/bench_r50_l1u.json | cut -c1-200
