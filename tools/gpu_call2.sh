#!/bin/bash
# round-2 GPU call 2: f32 parity plan + exact top-k + full GPU suite
set -x
export TMPDIR=/tmp
O=gpurun_out/c2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_f32_plan.py -q -x > $O/pytest_f32.log 2>&1; echo "rc=$?" >> $O/pytest_f32.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "det_select" > $O/pytest_topk.log 2>&1; echo "rc=$?" >> $O/pytest_topk.log
timeout 900 python tools/parity_baseline.py --depth 50 --batch 4 --precision f32 --out $O/parity_r50_b4_f32.json > $O/parity_f32.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 15 $O/pytest_f32.log; tail -n 5 $O/pytest_topk.log; tail -n 40 $O/parity_f32.log; tail -n 15 $O/pytest_all.log
