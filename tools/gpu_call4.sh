#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/c4
mkdir -p $O
for b in 1 2 8; do timeout 300 python bench.py --batch $b --no-cpu-baseline --breakdown $O/breakdown_b$b.txt > $O/bench_b$b.json 2> $O/bench_b$b.err; done
timeout 300 python bench.py --config vis > $O/bench_vis.json 2> $O/bench_vis.err
timeout 300 python bench.py --config train > $O/bench_train.json 2> $O/bench_train.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-graph --steps 20 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*stats*" | head; 
for b in 1 2 8; do cut -c1-260 $O/bench_b$b.json; done; cut -c1-300 $O/bench_vis.json; cut -c1-300 $O/bench_train.json
