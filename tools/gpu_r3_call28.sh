#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c28_wgrad.txt
timeout 300 python bench.py --config train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-330 | tee gpurun_out/r3c28_train.json
