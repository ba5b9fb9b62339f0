#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/c6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "mask_assemble or nms or det_select or rle" > $O/pytest_k.log 2>&1; echo "rc=$?" >> $O/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_benchmark_variant.py tests/test_gpu_vis.py tests/test_gpu_api.py -q -x > $O/pytest_e.log 2>&1; echo "rc=$?" >> $O/pytest_e.log
timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown.txt > $O/bench.json 2> $O/bench.err
SIPMASK_FUSED_MASKS=0 timeout 300 python bench.py --no-cpu-baseline > $O/bench_nofuse.json 2> $O/bench_nofuse.err
timeout 300 python bench.py --config vis > $O/bench_vis.json 2> $O/bench_vis.err
tail -n 12 $O/pytest_k.log $O/pytest_e.log; cut -c1-200 $O/bench.json $O/bench_nofuse.json $O/bench_vis.json; tail -n 3 $O/bench.err; grep -n "up:\|det_select\|nms\|mask_assemble\|gn:\|sum" $O/breakdown.txt
