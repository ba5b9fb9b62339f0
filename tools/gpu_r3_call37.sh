#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in 2 1; do for sl in 2 3; do SIPMASK_VIS_LANES=$l SIPMASK_VIS_SLOTS=$sl timeout 300 python bench.py --config vis --steps 12 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes', $l, 'slots', $sl, d['value'], d['ms_per_step'])"; done; done
