#!/bin/bash
mkdir -p gpurun_out/prof3; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_vis.py -x -q -m gpu 2>&1 | tail -2
timeout 400 python bench.py --config vis > gpurun_out/prof3/bench_vis.json 2> gpurun_out/prof3/bench_vis.err; cut -c1-300 gpurun_out/prof3/bench_vis.json; python -c "
import json; d=json.loads([l for l in open('gpurun_out/prof3/bench_vis.json') if l.startswith('{')][-1]); print(d['roofline']); print(d['config']['workload'])"
