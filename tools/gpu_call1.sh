#!/bin/bash
# round-2 GPU call 1: validate the merged grouped-tower work + measure bf16 parity at the BASELINE shape
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/c1
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "grouped or loader_variants or groupnorm_statistics" > $O/pytest_wip.log 2>&1; echo "rc=$?" >> $O/pytest_wip.log
timeout 300 python tools/conv_bench.py --only tower,cls_cof,fpn.out0 --variants 0,0x00040000,0x00400000,0x00440000 > $O/conv_bench.log 2>&1
SIPMASK_GROUPED_TOWERS=1 timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_api.py -q -x > $O/pytest_grouped.log 2>&1; echo "rc=$?" >> $O/pytest_grouped.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
SIPMASK_GROUPED_TOWERS=1 timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown_grouped.txt > $O/bench_grouped.json 2> $O/bench_grouped.err
SIPMASK_GROUPED_TOWERS=1 SIPMASK_CONV_DEBUG_FLAGS=0x00040000 timeout 300 python bench.py --no-cpu-baseline > $O/bench_grouped_hand.json 2> $O/bench_grouped_hand.err
timeout 900 python tools/parity_baseline.py --depth 50 --batch 4 --precision bf16 --out $O/parity_r50_b4_bf16.json > $O/parity_bf16.log 2>&1
tail -3 $O/pytest_wip.log $O/pytest_grouped.log; cat $O/conv_bench.log | tail -5; cat $O/bench_base.json $O/bench_grouped.json $O/bench_grouped_hand.json | cut -c1-300; tail -30 $O/parity_bf16.log
