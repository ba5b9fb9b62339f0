#!/bin/bash
mkdir -p gpurun_out/prof3; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof3; cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --no-cpu-baseline --lanes 1 --in-flight 1 > $OUT/bench_r50_lanes1.json 2>/dev/null
timeout 300 python $R/bench.py --in-flight 1 > $OUT/bench_r50_inflight1.json 2>/dev/null
timeout 300 python $R/bench.py --no-cpu-baseline --in-flight 3 > $OUT/bench_r50_inflight3.json 2>/dev/null
for f in lanes1 inflight1 inflight3; do cut -c1-160 $OUT/bench_r50_$f.json; done
