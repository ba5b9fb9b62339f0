#!/bin/bash
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for u in 0 1 0 1; do SIPMASK_PIPE_UNIFORM=$u timeout 300 python $R/bench.py --no-cpu-baseline --steps 60 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('uniform patch tiles in slots', $u, d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'])"; done
