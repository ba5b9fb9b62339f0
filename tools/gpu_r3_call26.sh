#!/bin/bash
# round 3, call 26: wgrad_direct with the XCD-aware block order (all blocks of a split-K slice on one XCD)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_rows.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r3c26_pytest.log 2>&1; tail -3 gpurun_out/r3c26_pytest.log
timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c26_wgrad.txt
timeout 300 python bench.py --config train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-330 | tee gpurun_out/r3c26_train.json
