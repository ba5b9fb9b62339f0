#!/bin/bash
mkdir -p gpurun_out/prof3; cd $GRAFT_REPO_ROOT
timeout 300 python tools/parity_baseline.py --plan pipelined --precision bf16 --out gpurun_out/prof3/parity_r50_b4_bf16_pipelined.json 2>&1 | tail -4 | cut -c1-300
