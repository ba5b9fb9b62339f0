#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/c9
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -q -x -k "deform or groupnorm_statistics or head" > $O/pytest_k.log 2>&1; echo "rc=$?" >> $O/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_baseline_shape.py -q -x > $O/pytest_e.log 2>&1; echo "rc=$?" >> $O/pytest_e.log
timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown.txt > $O/bench.json 2> $O/bench.err
SIPMASK_CONV_DEBUG_FLAGS=0x00008000 timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown_d128.txt > $O/bench_d128.json 2> $O/bench_d128.err
tail -n 6 $O/pytest_k.log $O/pytest_e.log; cut -c1-200 $O/bench.json $O/bench_d128.json; grep -n "feat_align" $O/breakdown.txt $O/breakdown_d128.txt
