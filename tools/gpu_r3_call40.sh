#!/bin/bash
# round 3, call 40: steps in flight (engine.PipelinedPlan): correctness, repeated, then the bench line
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "pipelined" 2>&1 | tail -1; done
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_baseline_shape.py -x -q -m gpu 2>&1 | tail -2
for sub in 0 1; do for n in 2 3; do
SIPMASK_PIPE_SUBPLAN=$sub timeout 300 python bench.py --in-flight $n --steps 60 --warmup 6 --no-cpu-baseline 2>gpurun_out/r3c40_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('subplan-style slots', $sub, 'in flight', $n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_launch'], d['config']['detections_per_image'])"
done; done
timeout 300 python bench.py --in-flight 1 --steps 40 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in flight 1', d['value'], d['ms_per_step'], d['roofline']['frac'])"
tail -3 gpurun_out/r3c40_err.txt
