#!/bin/bash
# round 3, GPU call 1: full GPU suite after the fixed-point GroupNorm statistics + bucket-resident gradients, the
# timed plan's parity at the BASELINE shape, one bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 --deselect tests/test_gpu_baseline_shape.py > gpurun_out/r3c1_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3c1_pytest.log
tail -30 gpurun_out/r3c1_pytest.log
timeout 600 python tools/parity_baseline.py --plan subbatch --precision bf16 --out gpurun_out/r3c1_parity_subbatch_bf16.json > gpurun_out/r3c1_parity.log 2>&1
tail -25 gpurun_out/r3c1_parity.log
timeout 600 python bench.py --steps 20 --warmup 5 --breakdown gpurun_out/r3c1_breakdown.txt > gpurun_out/r3c1_bench.json 2> gpurun_out/r3c1_bench.err
tail -c 3000 gpurun_out/r3c1_bench.json; tail -5 gpurun_out/r3c1_bench.err
