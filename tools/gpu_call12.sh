#!/bin/bash
# A/B round: relu-copy P7, patch cout rule, GN block height, split-K in sub-plans, per-sub-plan graphs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call12
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q -m gpu > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10"
for rep in 1 2; do
  $B --breakdown $OUT/bd_default_$rep.txt > $OUT/default_$rep.json 2>/dev/null
  SIPMASK_SPLIT_K=0 $B > $OUT/nosplitk_$rep.json 2>/dev/null
  $B --sub-graphs 1 > $OUT/subgraphs1_$rep.json 2>$OUT/subgraphs1_$rep.err
  $B --sub-graphs 2 > $OUT/subgraphs2_$rep.json 2>/dev/null
  SIPMASK_RELU_COPY_P7=0 $B > $OUT/norelucopy_$rep.json 2>/dev/null
done
for f in $OUT/*.json; do echo $(basename $f) $(python -c "import json,sys;l=[x for x in open('$f') if x.startswith('{')];j=json.loads(l[-1]) if l else {};print(j.get('value'),j.get('ms_per_step'))"); done
grep -E "gn:|fpn.p7|relu:p6|sip_mask_lat " $OUT/bd_default_1.txt
