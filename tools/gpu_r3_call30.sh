#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_rows.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r3c30_pytest.log 2>&1; tail -3 gpurun_out/r3c30_pytest.log
timeout 300 python tools/deform_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c30_deform_bwd.txt
timeout 300 python bench.py --config train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-330 | tee gpurun_out/r3c30_train.json
