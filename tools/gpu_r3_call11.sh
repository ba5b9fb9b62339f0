#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 300 python tools/patch_ring_bench.py > gpurun_out/r3c11_ring.txt 2>&1; cat gpurun_out/r3c11_ring.txt
