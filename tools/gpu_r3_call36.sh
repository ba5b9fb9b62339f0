#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in 4 8 16; do SIPMASK_VIS_CLIPS=$c timeout 300 python bench.py --config vis --steps 12 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('clips per step', $c, d['value'], d['ms_per_step'])"; done
