#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/c8
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "split_k or igemm_vs_torch or loader_variants" > $O/pytest_k.log 2>&1; echo "rc=$?" >> $O/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_vis.py tests/test_gpu_benchmark_variant.py -q -x > $O/pytest_e.log 2>&1; echo "rc=$?" >> $O/pytest_e.log
timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown.txt > $O/bench.json 2> $O/bench.err
SIPMASK_SPLIT_K=0 timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown_nosplit.txt > $O/bench_nosplit.json 2> $O/bench_nosplit.err
timeout 300 python bench.py --config vis > $O/bench_vis.json 2> $O/bench_vis.err
SIPMASK_SPLIT_K=0 timeout 300 python bench.py --config vis > $O/bench_vis_nosplit.json 2> $O/bench_vis_nosplit.err
timeout 300 python bench.py --config r101 --no-cpu-baseline > $O/bench_r101.json 2> $O/bench_r101.err
tail -n 6 $O/pytest_k.log $O/pytest_e.log; cut -c1-200 $O/bench.json $O/bench_nosplit.json $O/bench_vis.json $O/bench_vis_nosplit.json $O/bench_r101.json
grep -n "layer3.1\|layer4.1\|fpn" $O/breakdown.txt; grep -n "layer3.1\|layer4.1\|fpn" $O/breakdown_nosplit.txt
