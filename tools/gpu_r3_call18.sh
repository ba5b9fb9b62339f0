#!/bin/bash
# round 3, call 18: full GPU suite + the round's evidence (tools/profile_round3.sh)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r3c18_pytest.log 2>&1; tail -3 gpurun_out/r3c18_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round3.sh 2>&1 | tail -12
