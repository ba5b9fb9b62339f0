#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 400 python tools/patch_clock_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c21_clock.txt
