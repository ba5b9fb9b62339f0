#!/bin/bash
# fused bottleneck tails: parity, then A/B of SIPMASK_FUSE_BOTTLENECK = 0 / 1 / 2 with per-step breakdowns
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call13
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bottleneck or relu_bf16" > $OUT/pytest_k.log 2>&1
tail -3 $OUT/pytest_k.log
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu > $OUT/pytest_e.log 2>&1
tail -3 $OUT/pytest_e.log
B="timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10"
for rep in 1 2; do
  for m in 0 1 2; do
    SIPMASK_FUSE_BOTTLENECK=$m $B --breakdown $OUT/bd_fuse${m}_$rep.txt > $OUT/fuse${m}_$rep.json 2>$OUT/fuse${m}_$rep.err
  done
done
for f in $OUT/*.json; do echo $(basename $f) $(python -c "import json,sys;l=[x for x in open('$f') if x.startswith('{')];j=json.loads(l[-1]) if l else {};print(j.get('value'),j.get('ms_per_step'))"); done
grep -E "layer1\.|layer2\.0" $OUT/bd_fuse2_1.txt | head -20
grep -E "layer1\.1|layer2\.1" $OUT/bd_fuse0_1.txt | head
