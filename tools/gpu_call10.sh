#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/c10
mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline --breakdown $O/breakdown.txt > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --lanes 1 > $O/bench_l1.json 2> $O/bench_l1.err
timeout 300 python bench.py --no-cpu-baseline --config r101 > $O/bench_r101.json 2> $O/bench_r101.err
timeout 300 python bench.py --no-cpu-baseline --precision f32 > $O/bench_f32.json 2> $O/bench_f32.err
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_engine.py -q -x > $O/pytest_e.log 2>&1; echo "rc=$?" >> $O/pytest_e.log
cut -c1-200 $O/bench.json $O/bench_l1.json $O/bench_r101.json $O/bench_f32.json; tail -n 5 $O/bench.err $O/pytest_e.log
