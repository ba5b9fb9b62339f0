#!/bin/bash
# Last GPU call of a round, ordered by priority: full GPU test suite, rocprofv3 stats of the dominant kernel alone,
# the full bench line (with cpu_baseline), rocprofv3 stats of one eager step.  Outputs under gpurun_out/final/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
timeout 80 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -3 $OUT/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 5 40 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tower -- \
  python $R/bench.py --tower-only 50 > $OUT/tower.log 2>&1
tail -1 $OUT/tower.log
timeout 90 python $R/bench.py --steps 20 --warmup 5 --breakdown $OUT/step_breakdown.txt > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -1 $OUT/bench_full.json
timeout -k 5 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step -- \
  python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $OUT/step.log 2>&1
find $OUT -name "*stats*.csv" | head
