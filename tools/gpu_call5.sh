#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/c5
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "loader_variants or nms or groupnorm or det_select" > $O/pytest_k.log 2>&1; echo "rc=$?" >> $O/pytest_k.log
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_benchmark_variant.py -q -x > $O/pytest_e.log 2>&1; echo "rc=$?" >> $O/pytest_e.log
timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown_base.txt > $O/bench_base.json 2> $O/bench_base.err
SIPMASK_CONV_DEBUG_FLAGS=0x00020000 timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown_respf.txt > $O/bench_respf.json 2> $O/bench_respf.err
timeout 300 python tools/conv_bench.py --only l1.conv3,l2.conv3,l3.conv3 --variants 0,0x00020000 > $O/conv_bench.log 2>&1
tail -n 4 $O/pytest_k.log $O/pytest_e.log; cut -c1-200 $O/bench_base.json $O/bench_respf.json; cat $O/conv_bench.log; grep -n "gn:\|nms\|det_select\|mask_assemble\|conv3 " $O/breakdown_base.txt | head -40
