#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python tools/pipeline_steps_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c39_pipeline.txt
