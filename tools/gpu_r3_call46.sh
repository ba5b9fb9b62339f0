#!/bin/bash
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for fl in 0 0x04000000 0x00010000 0x04010000; do SIPMASK_CONV_DEBUG_FLAGS=$fl timeout 300 python $R/bench.py --no-cpu-baseline --steps 60 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags', '$fl', d['value'], d['ms_per_step'])"; done
