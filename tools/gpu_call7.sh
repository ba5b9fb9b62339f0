#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/c7
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_benchmark_variant.py tests/test_gpu_vis.py -q -x -k "nms or det_select or pairs or benchmark or vis" > $O/pytest_k.log 2>&1; echo "rc=$?" >> $O/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_baseline_shape.py -q -x > $O/pytest_e.log 2>&1; echo "rc=$?" >> $O/pytest_e.log
timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown.txt > $O/bench.json 2> $O/bench.err
tail -n 6 $O/pytest_k.log $O/pytest_e.log; cut -c1-200 $O/bench.json; grep -n "det_select\|nms\|mask_assemble\|sum" $O/breakdown.txt
