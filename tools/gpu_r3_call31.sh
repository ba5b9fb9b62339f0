#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu > gpurun_out/r3c31_pytest.log 2>&1; tail -3 gpurun_out/r3c31_pytest.log
for i in 1 2; do timeout 200 python bench.py --steps 40 --no-cpu-baseline --breakdown gpurun_out/r3c31_breakdown.txt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
grep -E "tail|stem|maxpool|sum" gpurun_out/r3c31_breakdown.txt
