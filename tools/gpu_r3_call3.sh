#!/bin/bash
# round 3, GPU call 3: full suite after the experiments split + binary16 (x3) kernels, x3 parity at the BASELINE shape,
# bench lines of the bf16 and head_x3 plans
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_x3.py -m gpu -q --maxfail=10 > gpurun_out/r3c3_pytest_x3.log 2>&1
echo "pytest x3 rc $?" >> gpurun_out/r3c3_pytest_x3.log
tail -40 gpurun_out/r3c3_pytest_x3.log
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 --deselect tests/test_gpu_x3.py > gpurun_out/r3c3_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3c3_pytest.log
tail -30 gpurun_out/r3c3_pytest.log
timeout 600 python tools/parity_baseline.py --plan subbatch --precision head_x3 --out gpurun_out/r3c3_parity_subbatch_x3.json > gpurun_out/r3c3_parity_x3.log 2>&1
tail -28 gpurun_out/r3c3_parity_x3.log
timeout 600 python bench.py --steps 20 --warmup 5 --precision head_x3 --breakdown gpurun_out/r3c3_breakdown_x3.txt > gpurun_out/r3c3_bench_x3.json 2> gpurun_out/r3c3_bench_x3.err
tail -c 2500 gpurun_out/r3c3_bench_x3.json; tail -5 gpurun_out/r3c3_bench_x3.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3c3_bench_bf16.json 2> gpurun_out/r3c3_bench_bf16.err
tail -c 1200 gpurun_out/r3c3_bench_bf16.json; tail -3 gpurun_out/r3c3_bench_bf16.err
