#!/bin/bash
# round 3, final: full GPU suite + smoke + the default bench line on the final tree
mkdir -p gpurun_out/prof3; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof3; cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r3c49_pytest.log 2>&1; tail -2 gpurun_out/r3c49_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py > $OUT/bench_r50.json 2> $OUT/bench_r50.err; cut -c1-330 $OUT/bench_r50.json
