#!/bin/bash
# round 3, call 13: pipelined VIS clips, DeformConv argument range, VIS bench
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vis.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r3c13_pytest.log 2>&1; tail -5 gpurun_out/r3c13_pytest.log
timeout 300 python bench.py --config vis --steps 20 > gpurun_out/r3c13_vis.json 2> gpurun_out/r3c13_vis.err; cut -c1-400 gpurun_out/r3c13_vis.json; tail -3 gpurun_out/r3c13_vis.err
timeout 300 python bench.py --config vis --steps 20 --no-cpu-baseline 2>/dev/null | cut -c1-200
