#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/c11
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_baseline_shape.py tests/test_gpu_api.py tests/test_gpu_vis.py tests/test_gpu_benchmark_variant.py -q -x > $O/pytest_e.log 2>&1; echo "rc=$?" >> $O/pytest_e.log
timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown.txt > $O/bench.json 2> $O/bench.err
SIPMASK_PATCH_CONV=0 timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown_nopatch.txt > $O/bench_nopatch.json 2> $O/bench_nopatch.err
timeout 300 python bench.py --no-cpu-baseline --lanes 1 > $O/bench_l1.json 2> $O/bench_l1.err
SIPMASK_PATCH_CONV=0 timeout 300 python bench.py --no-cpu-baseline --lanes 1 > $O/bench_l1_nopatch.json 2> $O/bench_l1_nopatch.err
tail -n 6 $O/pytest_e.log; cut -c1-200 $O/bench.json $O/bench_nopatch.json $O/bench_l1.json $O/bench_l1_nopatch.json; tail -n 3 $O/bench.err; grep -n "tower\|cls_cof\|fpn.out" $O/breakdown.txt $O/breakdown_nopatch.txt
