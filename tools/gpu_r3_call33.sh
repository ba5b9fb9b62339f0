#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_rows.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r3c33_pytest.log 2>&1; tail -3 gpurun_out/r3c33_pytest.log
for i in 1 2; do timeout 300 python bench.py --config train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
