#!/bin/bash
# round 3, call 14: is test_detector_forward_train_vs_oracle flaky?  + training step kernel profile
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "detector_forward_train_vs_oracle" 2>&1 | grep -E "passed|failed|AssertionError: \[" | tr '\n' ' '; echo
done | tee gpurun_out/r3c14_flaky.txt
timeout 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r3c14_pytest.log 2>&1; tail -3 gpurun_out/r3c14_pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3c14_train -- python $GRAFT_REPO_ROOT/bench.py --config train --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3c14_train.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3c14_train.err
cd $GRAFT_REPO_ROOT; cut -c1-300 gpurun_out/r3c14_train.json
f=$(find gpurun_out/r3c14_train -name "*kernel_stats.csv" | head -1); head -40 "$f" | cut -c1-200
