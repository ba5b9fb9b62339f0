#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_baseline_shape.py -x -q -m gpu -k "pipelined" 2>&1 | tail -3
