#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 300 python tools/patch_cu_scaling_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c20_cu_scaling.txt
