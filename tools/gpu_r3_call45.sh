#!/bin/bash
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for n in 3 4 6; do timeout 300 python $R/bench.py --no-cpu-baseline --in-flight $n --steps 60 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in flight', $n, d['value'], d['ms_per_step'])"; done
